// graph_exec.hip -- graph D as a native executor behind the C ABI: emd_graph_create / _workspace_bytes / _run / _destroy
// (last row).  Replaces architecture() of machine_learning/denoiser.py:58-398 for a host that is not Python: the
// layer table in the reference's variable-creation order (so that every weight keeps its TensorFlow name), the folding of the
// inference batch norms into per-channel affines (float64), the bf16 hi / lo weight packing, and the launch sequence of the
// library's own entry points over a caller-provided device workspace.  Same kernel choices as the Python engine
// (emdenoise.denoiser.DenoiserEngine), single stream: results are bit-identical to it (tests/test_graph_exec_gpu.py).
//
// Ownership: emd_graph_create uploads the prepared parameters into device memory it allocates and emd_graph_destroy frees (the
// one explicit handle of the ABI); activations live in the caller's workspace (emd_graph_workspace_bytes), nothing else is
// allocated, no global state.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "graph_common.hpp"

namespace {

constexpr int F0 = 64, F1 = 128, F2 = 256, F3 = 728, F4 = 728, AF = 728, AOUT = 256, NEXTRA = 11;
constexpr double BN_EPS = 1e-3;

enum Kind { SEP, CONV, DECONV, BNONLY };

struct LayerDecl {
    std::string key;
    Kind kind;
    int cin, cout, k = 1, stride = 1, rate = 1;
    std::string wname = "weights", bname = "biases";   // slim names; tf.layers (variant D'): kernel / bias
    std::string scope;             // conv / separable-conv / transposed-conv scope
    std::vector<std::string> bn;   // batch norms applied after it, in order
    std::string extra_bn;          // the ASPP rate branches' second batch_then_activ
};

// tf.variable_scope default-name uniquifier inside scope 'nn' (denoiser.py:514)
struct Scope {
    std::map<std::string, int> n;
    std::string operator()(const std::string& base) {
        const int k = n[base]++;
        return k == 0 ? "nn/" + base : "nn/" + base + "_" + std::to_string(k);
    }
};

// every parameterised layer of architecture() in creation order (mirror of emdenoise.denoiser.declare_layers): twin = false graph D
// (machine_learning/denoiser.py:248-398, slim layers), twin = true graph D' (misc_py/denoiser-multi-gpu.py:200-540 with phase=False:
// tf.layers convs named conv2d_k / conv2d_transpose_k with kernel / bias, named ASPP convs, dense dilated 3x3 ASPP branches, a real
// image-level branch)
std::vector<LayerDecl> declare_layers(bool twin) {
    Scope sc;
    std::vector<LayerDecl> L;
    auto sep = [&](const std::string& key, int cin, int cout, int stride = 1, int rate = 1, bool extra = false) {
        LayerDecl d{key, SEP, cin, cout, 3, stride, rate};
        d.scope = sc("SeparableConv2d");
        d.bn = {d.scope + "/BatchNorm", sc("BatchNorm")};   // normalizer_fn inside the layer's scope (:123), then batch_then_activ (:134)
        if (extra) d.extra_bn = sc("BatchNorm");
        L.push_back(d);
    };
    auto conv = [&](const std::string& key, int cin, int cout, int k = 1, int stride = 1, int rate = 1, const char* name = nullptr,
                    bool bn = true) {
        LayerDecl d{key, CONV, cin, cout, k, stride, rate};
        d.scope = twin ? (name ? std::string("nn/") + name : sc("conv2d")) : sc("Conv");
        if (twin) { d.wname = "kernel"; d.bname = "bias"; }
        if (bn) d.bn = {sc("BatchNorm")};
        L.push_back(d);
    };
    auto deconv = [&](const std::string& key, int cin, int cout) {
        LayerDecl d{key, DECONV, cin, cout, 3, 2, 1};
        d.scope = twin ? sc("conv2d_transpose") : sc("Conv2d_transpose");
        if (twin) { d.wname = "kernel"; d.bname = "bias"; }
        d.bn = {sc("BatchNorm")};
        L.push_back(d);
    };
    sep("cnn0", 1, F0); sep("cnn0_last", F0, F0); sep("cnn0_strided", F0, F1, 2);
    conv("residual0", 1, F1, 1, 2);
    sep("cnn1", F1, F1); sep("cnn1_last", F1, F1); sep("cnn1_strided", F1, F1, 2);
    conv("residual1", F1, F1, 1, 2);
    sep("cnn2", F1, F2); sep("cnn2_last", F2, F2); sep("cnn2_strided", F2, F2, 2);
    conv("residual2", F1, F2, 1, 2);
    sep("cnn3", F2, F3); sep("cnn3_last", F3, F3); sep("cnn3_strided", F3, F3, 2);
    conv("residual3", F2, F3, 1, 2);
    sep("cnn4_a", F3, F4); sep("cnn4_b", F4, F4); sep("cnn4_last", F4, F4);
    for (int i = 0; i < NEXTRA; ++i)
        for (int j = 0; j < 3; ++j) sep("middle" + std::to_string(i) + "_" + std::to_string(j), F4, F4);
    if (!twin) {
        conv("aspp_conv1x1", F4, AF);
        sep("aspp_small", F4, AF, 1, 6, true); sep("aspp_medium", F4, AF, 1, 12, true); sep("aspp_large", F4, AF, 1, 18, true);
        LayerDecl d{"aspp_pooling_bn", BNONLY, F4, F4};
        d.bn = {sc("BatchNorm")};   // :199-200
        L.push_back(d);
        conv("aspp_reduce", 5 * AF, AOUT);
    } else {   // denoiser-multi-gpu.py:291-361
        conv("aspp_conv1x1", F4, AF, 1, 1, 1, "1x1");
        conv("aspp_small", F4, AF, 3, 1, 6, "lowRate");
        conv("aspp_medium", F4, AF, 3, 1, 12, "mediumRate");
        conv("aspp_large", F4, AF, 3, 1, 18, "highRate");
        conv("aspp_image_conv", F4, AF, 1, 1, 1, "imageLevel", false);   // conv -> resize -> BN -> relu6
        LayerDecl d{"aspp_pooling_bn", BNONLY, AF, AF};
        d.bn = {sc("BatchNorm")};
        L.push_back(d);
        conv("aspp_reduce", 5 * AF, AOUT, 1, 1, 1, "pellet");
    }
    sep("deconv2_a", AOUT + F1, F2); sep("deconv2_b", F2, F2);
    conv("residual2_d", AOUT + F1, F2);
    deconv("deconv2to1", F2, F2);
    sep("deconv1_a", F2 + F1, F1); sep("deconv1_b", F1, F1);
    conv("residual1_d", F2 + F1, F1);
    deconv("deconv1to0", F1, F1);
    sep("deconv0_a", F1, F0); sep("deconv0_b", F0, F0);
    conv("residual0_d", F1, F0);
    conv("deconv_final", F0, 1, 3);   // :387 -- kernel_size defaults to 3
    return L;
}


using emd::gx::Packed;

struct LayerParams {
    LayerDecl d;
    float* dw = nullptr;                          // [9][Cin]
    Packed pw;                                    // pointwise / 1x1 / 3x3 conv weights
    Packed phase[4];                              // transposed conv
    float *scale = nullptr, *shift = nullptr;     // folded affine(s)
    float *scale2 = nullptr, *shift2 = nullptr;   // the extra batch norm of the ASPP rate branches
    float *w9 = nullptr, *a = nullptr;            // layers fed by the 1-channel image: depthwise taps, outer-product vector
    float* wfin = nullptr;                        // final 3x3 -> 1 conv: [9][Cin]
    float scale_f = 1.f, shift_f = 0.f;
};

}  // namespace

struct emd_graph {   // the handle behind emd_graph_t
    std::map<std::string, LayerParams> P;
    std::vector<void*> allocs;
    float *unit4 = nullptr, *zero4 = nullptr;
    std::string error;
    // the two side streams of the 1/16-resolution flow (Run::middle_two_streams) and their fork / join events; made on first use
    bool twin = false;          // variant 1: graph D' (the training twin's inference graph)
    emd::gx::XGraph* x = nullptr;   // variant 2: graph X (graph_exec_x.hip)
    emd::gx::GGraph* gen = nullptr; // variant 3: graph G's generator (graph_exec_g.hip)
    bool two_streams = false;   // emd_graph_set_two_streams
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    bool streams_failed = false;   // creation failed once: later runs go straight to the single-stream sequence (no retry, no second leak)
    void drop_streams() {
        for (int h = 0; h < 2; ++h) {
            if (join[h]) (void)hipEventDestroy(join[h]);
            if (side[h]) (void)hipStreamDestroy(side[h]);
            join[h] = nullptr; side[h] = nullptr;
        }
        if (fork) (void)hipEventDestroy(fork);
        fork = nullptr;
    }
    bool streams_ok() {
        if (streams_failed) return false;
        if (side[0] && side[1]) return true;
        bool ok = hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess;
        for (int h = 0; h < 2 && ok; ++h)
            ok = hipStreamCreateWithFlags(&side[h], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&join[h], hipEventDisableTiming) == hipSuccess;
        if (!ok) {   // single-stream launch sequence instead; whatever was created goes
            drop_streams();
            streams_failed = true;
        }
        return ok;
    }
};

namespace {

using emd::gx::WeightMap;
using emd::gx::fetch;
using emd::gx::Arena;
using emd::gx::T4;

template <typename T>
T* upload(emd_graph* g, const T* host, size_t n) { return emd::gx::upload(g->allocs, host, n); }
float* upload_f(emd_graph* g, const std::vector<double>& v) { return emd::gx::upload_f(g->allocs, v); }
bool pack(emd_graph* g, const float* w, int taps, int cin, int cout, int cout_major, Packed* out) {
    return emd::gx::pack(g->allocs, w, taps, cin, cout, cout_major, out);
}
bool bn_affine(const WeightMap& w, const std::string& scope, int C, std::vector<double>* gs, std::vector<double>* hs, std::string* err) {
    return emd::gx::bn_affine(w, scope, C, BN_EPS, gs, hs, err);
}

// bias + the layer's consecutive batch norms -> one affine (scale, shift)
bool fold(const WeightMap& w, const LayerDecl& d, const float* bias, std::vector<double>* s, std::vector<double>* t, std::string* err) {
    const int C = d.cout;
    s->assign(C, 1.0);
    t->assign(C, 0.0);
    if (bias)
        for (int c = 0; c < C; ++c) (*t)[c] = bias[c];
    for (const std::string& scope : d.bn) {
        std::vector<double> gs, hs;
        if (!bn_affine(w, scope, C, &gs, &hs, err)) return false;
        for (int c = 0; c < C; ++c) {
            (*s)[c] *= gs[c];
            (*t)[c] = (*t)[c] * gs[c] + hs[c];
        }
    }
    return true;
}

struct Run {
    emd_graph* g;
    Arena* ar;
    hipStream_t st;
    bool dry;
    int rc = EMD_OK;

    T4 E(int B, int H, int W, int C) {
        T4 t;
        t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C;
        t.buf = static_cast<float*>(ar->alloc((size_t)B * H * W * C * 4));
        if (!t.buf && rc == EMD_OK) rc = emd::fail(EMD_E_INVALID, "emd_graph_run: workspace too small");
        return t;
    }
    void* raw(size_t bytes) {
        void* p = ar->alloc(bytes);
        if (!p && rc == EMD_OK) rc = emd::fail(EMD_E_INVALID, "emd_graph_run: workspace too small");
        return p;
    }
    void free(T4& t) {
        ar->release(t.buf);
        t.buf = nullptr;
    }
    void call(int code) {
        if (code != EMD_OK && rc == EMD_OK) rc = code;
    }
    bool live() const { return !dry && rc == EMD_OK; }

    static bool split_gemm_ok(long npix, int cout, int ktot) {
        return cout >= 128 && ktot >= 512 && ((npix + 255) / 256) * ((cout + 127) / 128) >= 192;
    }
    static bool deconv_fused_ok(int B, int H, int W, int cin, int cout) { return emd_deconv3x3s2_fused_preferred(B, H, W, cin, cout) != 0; }

    // strided_conv_block (denoiser.py:110-136); out: optional destination (a concat slice); split_out: write a split32 tensor
    T4 sep(const std::string& key, const T4& x, const T4* out_opt, const T4* res, void** split_out = nullptr) {
        const LayerParams& p = g->P[key];
        const LayerDecl& d = p.d;
        const int Ho = (x.H + d.stride - 1) / d.stride, Wo = (x.W + d.stride - 1) / d.stride;
        const long M = (long)x.B * Ho * Wo;
        T4 out;
        const bool want_split = split_out != nullptr;
        const int ld_split = emd_split32_ld(d.cout);
        if (want_split) {
            *split_out = raw((size_t)M * ld_split * 4);
        } else {
            out = out_opt ? *out_opt : E(x.B, Ho, Wo, d.cout);
        }
        if (emd_sep3x3_fused_supported(x.H, x.W, d.cin, d.cout, d.stride, d.rate) && !(d.stride == 2 && want_split)) {
            if (live()) {
                if (d.stride == 2)
                    call(emd_sep3x3_fused_s2_f32(x.ptr(), x.ld, p.dw, p.pw.hi, p.pw.lo, p.scale, p.shift, p.scale2, p.shift2,
                                                 res ? res->ptr() : nullptr, res ? res->ld : 0, out.ptr(), out.ld, x.B, x.H, x.W, d.cin,
                                                 d.cout, EMD_ACT_RELU6, st));
                else if (want_split)
                    call(emd_sep3x3_fused_out_f32(x.ptr(), x.ld, p.dw, p.pw.hi, p.pw.lo, p.scale, p.shift, p.scale2, p.shift2,
                                                  res ? res->ptr() : nullptr, res ? res->ld : 0, *split_out, ld_split, x.B, x.H, x.W, d.cin,
                                                  d.cout, EMD_ACT_RELU6, st));
                else
                    call(emd_sep3x3_fused_f32(x.ptr(), x.ld, p.dw, p.pw.hi, p.pw.lo, p.scale, p.shift, p.scale2, p.shift2,
                                              res ? res->ptr() : nullptr, res ? res->ld : 0, out.ptr(), out.ld, x.B, x.H, x.W, d.cin, d.cout,
                                              EMD_ACT_RELU6, EMD_PREC_BF16X3, st));
            }
            return out;
        }
        if (emd_conv1x1_split32_supported(M, d.cin, d.cout)) {
            const int ldd = emd_split32_ld(d.cin);
            void* dsp = raw((size_t)M * ldd * 4);
            if (live()) {
                call(emd_dw3x3_split32_f32(x.ptr(), x.ld, p.dw, dsp, ldd, x.B, x.H, x.W, d.cin, d.stride, d.rate, st));
                if (want_split)
                    call(emd_conv1x1_split32_out_f32(dsp, ldd, p.pw.hi, p.pw.lo, p.scale, p.shift, p.scale2, p.shift2,
                                                     res ? res->ptr() : nullptr, res ? res->ld : 0, *split_out, ld_split, M, d.cin, d.cout,
                                                     EMD_ACT_RELU6, st));
                else
                    call(emd_conv1x1_split32_f32(dsp, ldd, p.pw.hi, p.pw.lo, p.scale, p.shift, p.scale2, p.shift2,
                                                 res ? res->ptr() : nullptr, res ? res->ld : 0, out.ptr(), out.ld, M, d.cin, d.cout,
                                                 EMD_ACT_RELU6, st));
            }
            ar->release(dsp);
            return out;
        }
        if (want_split && rc == EMD_OK) rc = emd::fail(EMD_E_UNSUPPORTED, "emd_graph_run: no split32-writing kernel for this layer shape");
        T4 tmp = E(x.B, Ho, Wo, d.cin);
        if (live()) {
            call(emd_dw3x3_f32(x.ptr(), x.ld, p.dw, tmp.ptr(), tmp.ld, x.B, x.H, x.W, d.cin, d.stride, d.rate, st));
            call(emd_conv1x1_f32(tmp.ptr(), tmp.ld, p.pw.hi, p.pw.lo, p.scale, p.shift, p.scale2, p.shift2, res ? res->ptr() : nullptr,
                                 res ? res->ld : 0, out.ptr(), out.ld, x.B, Ho, Wo, d.cin, d.cout, 1, EMD_ACT_RELU6, EMD_PREC_BF16X3, st));
        }
        free(tmp);
        return out;
    }

    // slim.conv2d(k = 1[, stride 2]) + bias + BN + relu6; xs: the input already in split32 form (shared by the ASPP branches)
    T4 conv1x1(const std::string& key, const T4& x, const T4* out_opt, const void* xs = nullptr, int ldxs = 0) {
        const LayerParams& p = g->P[key];
        const LayerDecl& d = p.d;
        const int Ho = (x.H + d.stride - 1) / d.stride, Wo = (x.W + d.stride - 1) / d.stride;
        T4 out = out_opt ? *out_opt : E(x.B, Ho, Wo, d.cout);
        if (!live()) return out;
        const int act = d.bn.empty() ? EMD_ACT_NONE : EMD_ACT_RELU6;   // conv + bias alone (D' image-level branch), or + BN + relu6
        // the pointwise split32 GEMM (16x16x32 MFMAs) serves a layer at every batch size or at none (another K-step summation order than
        // the register-staged kernel: image b of a batch must equal the image alone)
        if (xs && d.stride == 1 && emd_conv1x1_split32_supported(1L << 20, d.cin, d.cout))
            call(emd_conv1x1_split32_f32(xs, ldxs, p.pw.hi, p.pw.lo, p.scale, p.shift, nullptr, nullptr, nullptr, 0, out.ptr(), out.ld,
                                         (long)x.B * Ho * Wo, d.cin, d.cout, act, st));
        else
            call(emd_conv1x1_f32(x.ptr(), x.ld, p.pw.hi, p.pw.lo, p.scale, p.shift, nullptr, nullptr, nullptr, 0, out.ptr(), out.ld, x.B, x.H,
                                 x.W, d.cin, d.cout, d.stride, act, EMD_PREC_BF16X3, st));
        return out;
    }

    // tf.layers.conv2d(kernel_size = 3, dilation_rate, 'same') + bias + BN + relu6: D' dense ASPP rate branches
    // (misc_py/denoiser-multi-gpu.py:306-328); xs: the input already in split32 form, or NULL (converted here when the split32 GEMM pays)
    void conv3x3(const std::string& key, const T4& x, const T4& out, const void* xs, int ldxs) {
        const LayerParams& p = g->P[key];
        const LayerDecl& d = p.d;
        const long npix = (long)x.B * x.H * x.W;
        if (d.stride == 1 && split_gemm_ok(npix, d.cout, 9 * d.cin)) {
            void* tmp = nullptr;
            if (!xs) {
                ldxs = emd_split32_ld(d.cin);
                tmp = raw((size_t)npix * ldxs * 4);
                if (live()) call(emd_to_split32_f32(x.ptr(), x.ld, tmp, ldxs, npix, d.cin, st));
                xs = tmp;
            }
            if (live())
                call(emd_conv3x3_split32_f32(xs, ldxs, p.pw.hi, p.pw.lo, p.scale, p.shift, nullptr, nullptr, nullptr, 0, out.ptr(), out.ld, x.B,
                                             x.H, x.W, d.cin, d.cout, 1, d.rate, EMD_ACT_RELU6, 0, st));
            ar->release(tmp);
            return;
        }
        if (live())
            call(emd_conv3x3_f32(x.ptr(), x.ld, p.pw.hi, p.pw.lo, p.scale, p.shift, nullptr, nullptr, nullptr, 0, out.ptr(), out.ld, x.B, x.H, x.W,
                                 d.cin, d.cout, d.stride, d.rate, EMD_ACT_RELU6, EMD_PREC_BF16X3, st));
    }

    // slim.conv2d_transpose(k = 3, stride 2) + bias + BN + relu6 into `out`; x fp32, or xs a split32 tensor
    void deconv(const std::string& key, const T4* x, const void* xs, int B, int H, int W, const T4& out) {
        const LayerParams& p = g->P[key];
        const LayerDecl& d = p.d;
        const uint16_t* hi[4] = {p.phase[0].hi, p.phase[1].hi, p.phase[2].hi, p.phase[3].hi};
        const uint16_t* lo[4] = {p.phase[0].lo, p.phase[1].lo, p.phase[2].lo, p.phase[3].lo};
        const long npix = (long)B * H * W;
        const int ldx = emd_split32_ld(d.cin);
        const bool fused = deconv_fused_ok(B, H, W, d.cin, d.cout);
        if (fused || (d.cin >= 256 && split_gemm_ok(npix, d.cout, 4 * d.cin))) {
            void* tmp = nullptr;
            if (!xs) {
                tmp = raw((size_t)npix * ldx * 4);
                if (live()) call(emd_to_split32_f32(x->ptr(), x->ld, tmp, ldx, npix, d.cin, st));
                xs = tmp;
            }
            if (live()) {
                if (fused)
                    call(emd_deconv3x3s2_fused_split32_f32(xs, ldx, hi, lo, p.scale, p.shift, out.ptr(), out.ld, B, H, W, d.cin, d.cout,
                                                           EMD_ACT_RELU6, 0, st));
                else
                    call(emd_deconv3x3s2_split32_f32(xs, ldx, hi, lo, p.scale, p.shift, out.ptr(), out.ld, B, H, W, d.cin, d.cout,
                                                     EMD_ACT_RELU6, 0, st));
            }
            ar->release(tmp);
            return;
        }
        if (live())
            call(emd_deconv3x3s2_f32(x->ptr(), x->ld, hi, lo, p.scale, p.shift, out.ptr(), out.ld, B, H, W, d.cin, d.cout, EMD_ACT_RELU6,
                                     EMD_PREC_BF16X3, st));
    }

    // the decoder pair that reads the same tensor (denoiser.py:356-359, :368-371, :380-383)
    void sep_and_projection(const std::string& sk, const std::string& ck, const T4& x, T4* sep_out, T4* proj_out) {
        const LayerParams &ps = g->P[sk], &pc = g->P[ck];
        if (!ps.scale2 && emd_sep3x3_dual_preferred(x.H, x.W, x.C, ps.d.cout, pc.d.cout)) {
            *sep_out = E(x.B, x.H, x.W, ps.d.cout);
            *proj_out = E(x.B, x.H, x.W, pc.d.cout);
            if (live())
                call(emd_sep3x3_dual_f32(x.ptr(), x.ld, ps.dw, ps.pw.hi, ps.pw.lo, ps.scale, ps.shift, sep_out->ptr(), sep_out->ld, pc.pw.hi,
                                         pc.pw.lo, pc.scale, pc.shift, proj_out->ptr(), proj_out->ld, x.B, x.H, x.W, x.C, ps.d.cout, pc.d.cout,
                                         EMD_ACT_RELU6, st));
            return;
        }
        *proj_out = conv1x1(ck, x, nullptr);
        *sep_out = sep(sk, x, nullptr, nullptr);
    }

    // One residual block of the 1/16-resolution flow on *cur (a whole batch or a part of one): blk = -1 encoder 4 (:312-322, *cur is the
    // caller's tensor and stays), blk >= 0 middle block blk (:324-325, *cur came from this arena and is released).  last_out: where
    // the final block writes (a part of the caller's output tensor), or NULL.
    void middle_block(int blk, T4* cur, const T4* last_out) {
        const bool last = blk == NEXTRA - 1;
        const std::string k = blk < 0 ? "cnn4_" : "middle" + std::to_string(blk) + "_";
        T4 a = sep(blk < 0 ? k + "a" : k + "0", *cur, nullptr, nullptr);
        T4 b = sep(blk < 0 ? k + "b" : k + "1", a, nullptr, nullptr);
        free(a);
        T4 nxt = sep(blk < 0 ? k + "last" : k + "2", b, last ? last_out : nullptr, cur);
        free(b);
        if (blk >= 0) free(*cur);
        *cur = nxt;
    }

    // The same flow with the two halves of the batch on two streams, block by block (as emdenoise/streams.py TwoHalves: one half's
    // depthwise kernels share the chip with the other half's GEMMs; the images are independent, same bits).  Each half allocates from
    // its own sub-arena -- a buffer released on one stream is never handed to the other -- carved out of the caller's workspace.
    T4 middle_two_streams(const T4& x) {
        const int half = x.B / 2;
        T4 out = E(x.B, x.H, x.W, F4);
        auto part = [&](const T4& t, int h) {
            T4 v = t;
            v.B = half;
            if (v.buf) v.buf += (size_t)h * half * t.H * t.W * t.ld;
            return v;
        };
        // what one half needs: the same launch sequence against a measuring arena
        size_t need;
        {
            Arena m;
            m.measuring = true;
            Run sub{g, &m, nullptr, true};
            T4 c = part(x, 0);
            c.buf = reinterpret_cast<float*>(4096);
            T4 o = c;
            o.C = o.ld = F4;
            sub.middle_block(-1, &c, NEXTRA == 0 ? &o : nullptr);
            for (int i = 0; i < NEXTRA; ++i) sub.middle_block(i, &c, &o);
            need = m.peak + 256;
        }
        Arena sub[2];
        void* mem[2];
        for (int h = 0; h < 2; ++h) {
            mem[h] = raw(need);
            sub[h].measuring = dry;
            sub[h].base = static_cast<unsigned char*>(mem[h]);
            sub[h].cap = need;
        }
        if (rc != EMD_OK) return out;
        Arena* ar0 = ar;
        hipStream_t st0 = st;
        if (live()) {
            call(hipEventRecord(g->fork, st0) == hipSuccess ? EMD_OK : emd::fail(EMD_E_LAUNCH, "emd_graph_run: hipEventRecord failed"));
            for (int h = 0; h < 2; ++h)
                call(hipStreamWaitEvent(g->side[h], g->fork, 0) == hipSuccess ? EMD_OK : emd::fail(EMD_E_LAUNCH, "emd_graph_run: hipStreamWaitEvent failed"));
        }
        T4 cur[2] = {part(x, 0), part(x, 1)}, oh[2] = {part(out, 0), part(out, 1)};
        for (int blk = -1; blk < NEXTRA; ++blk)
            for (int h = 0; h < 2; ++h) {
                ar = &sub[h];
                st = dry ? st0 : g->side[h];
                middle_block(blk, &cur[h], &oh[h]);
            }
        ar = ar0;
        st = st0;
        if (!dry) {   // join even after an error: the main stream must not run ahead of launches already issued
            for (int h = 0; h < 2; ++h)
                if (hipEventRecord(g->join[h], g->side[h]) != hipSuccess || hipStreamWaitEvent(st0, g->join[h], 0) != hipSuccess)
                    call(emd::fail(EMD_E_LAUNCH, "emd_graph_run: stream join failed"));
        }
        for (int h = 0; h < 2; ++h) ar->release(mem[h]);
        return out;
    }

    // architecture() (denoiser.py:248-398) on x [B,S,S,1] -> y [B,S,S,1]
    void forward(const float* xin, float* yout, int B, int S) {
        const int S2 = S / 2, S4 = S / 4, S16 = S / 16;
        auto& P = g->P;
        // encoder 0 (:252-264): cnn0 = relu6(d * a + t) is an outer product of the 1-channel depthwise result d -- cnn0_last's patch
        // loader rebuilds it (emd_sep3x3_fused_gen_f32) where the fused kernel covers the shape
        T4 cnn0_last;
        if (emd_sep3x3_fused_supported(S, S, F0, F0, 1, 1)) {
            T4 d4 = E(B, S, S, 4);
            cnn0_last = E(B, S, S, F0);
            if (live()) {
                const LayerParams &pc = P["cnn0"], &pl = P["cnn0_last"];
                call(emd_cin1_f32(xin, pc.w9, g->unit4, g->zero4, d4.ptr(), d4.ld, B, S, S, 4, 1, 0, st));
                call(emd_sep3x3_fused_gen_f32(d4.ptr(), d4.ld, pc.a, pc.shift, EMD_ACT_RELU6, pl.dw, pl.pw.hi, pl.pw.lo, pl.scale, pl.shift, nullptr,
                                              nullptr, nullptr, 0, cnn0_last.ptr(), cnn0_last.ld, B, S, S, F0, F0, EMD_ACT_RELU6, EMD_PREC_BF16X3, 0, st));
            }
            free(d4);
        } else {
            T4 cnn0 = E(B, S, S, F0);
            if (live()) call(emd_cin1_f32(xin, P["cnn0"].w9, P["cnn0"].a, P["cnn0"].shift, cnn0.ptr(), cnn0.ld, B, S, S, F0, 1, 1, st));
            cnn0_last = sep("cnn0_last", cnn0, nullptr, nullptr);
            free(cnn0);
        }
        T4 residual0 = E(B, S2, S2, F1);
        if (live()) call(emd_cin1_f32(xin, nullptr, P["residual0"].a, P["residual0"].shift, residual0.ptr(), residual0.ld, B, S, S, F1, 2, 1, st));
        T4 concat1 = E(B, S2, S2, F2 + F1);
        T4 c1s = concat1.slice(F2, F1);
        T4 cnn0_strided = sep("cnn0_strided", cnn0_last, &c1s, &residual0);
        free(cnn0_last);
        free(residual0);
        // encoder 1 (:267-279)
        T4 residual1 = conv1x1("residual1", cnn0_strided, nullptr);
        T4 cnn1 = sep("cnn1", cnn0_strided, nullptr, nullptr);
        T4 cnn1_last = sep("cnn1_last", cnn1, nullptr, nullptr);
        T4 concat2 = E(B, S4, S4, AOUT + F1);
        T4 c2s = concat2.slice(AOUT, F1);
        T4 cnn1_strided = sep("cnn1_strided", cnn1_last, &c2s, &residual1);
        free(cnn1); free(cnn1_last); free(residual1);
        // encoder 2 (:282-294)
        T4 residual2 = conv1x1("residual2", cnn1_strided, nullptr);
        T4 cnn2 = sep("cnn2", cnn1_strided, nullptr, nullptr);
        T4 cnn2_last = sep("cnn2_last", cnn2, nullptr, nullptr);
        T4 cnn2_strided = sep("cnn2_strided", cnn2_last, nullptr, &residual2);
        free(cnn2); free(cnn2_last); free(residual2);
        // encoder 3 (:297-309)
        T4 residual3 = conv1x1("residual3", cnn2_strided, nullptr);
        T4 cnn3 = sep("cnn3", cnn2_strided, nullptr, nullptr);
        T4 cnn3_last = sep("cnn3_last", cnn3, nullptr, nullptr);
        T4 cnn3_strided = sep("cnn3_strided", cnn3_last, nullptr, &residual3);
        free(cnn2_strided); free(cnn3); free(cnn3_last); free(residual3);
        // encoder 4 (:312-322) and the middle flow (:324-325)
        T4 cur;
        if (g->two_streams && B % 2 == 0 && emd_conv1x1_split32_supported((long)(B / 2) * S16 * S16, F4, F4) && (dry || g->streams_ok())) {
            cur = middle_two_streams(cnn3_strided);
        } else {
            T4 x4 = cnn3_strided;
            middle_block(-1, &x4, nullptr);
            cur = x4;
            for (int i = 0; i < NEXTRA; ++i) middle_block(i, &cur, nullptr);
        }
        free(cnn3_strided);
        // ASPP (:152-216): the five branches write straight into their slices of the 3640-channel concat
        T4 cat = E(B, S16, S16, 5 * AF);
        void* curs = nullptr;
        const int ldcs = emd_split32_ld(AF);
        const long npix16 = (long)B * S16 * S16;
        if (emd_conv1x1_split32_supported(1L << 20, cur.C, AF) || (g->twin && split_gemm_ok(npix16, AF, 9 * cur.C))) {
            curs = raw((size_t)npix16 * ldcs * 4);
            if (live()) call(emd_to_split32_f32(cur.ptr(), cur.ld, curs, ldcs, npix16, cur.C, st));
        }
        T4 s0 = cat.slice(0, AF), s1 = cat.slice(AF, AF), s2 = cat.slice(2 * AF, AF), s3 = cat.slice(3 * AF, AF), s4 = cat.slice(4 * AF, AF);
        conv1x1("aspp_conv1x1", cur, &s0, curs, ldcs);
        if (!g->twin) {
            sep("aspp_small", cur, &s1, nullptr);
            sep("aspp_medium", cur, &s2, nullptr);
            sep("aspp_large", cur, &s3, nullptr);
            // :185-189 the pooled tensor is discarded; :199 "pooling" = an identity resize of the INPUT, then BN + relu6 (:200)
            if (live())
                call(emd_affine_relu6_f32(cur.ptr(), cur.ld, P["aspp_pooling_bn"].scale, P["aspp_pooling_bn"].shift, s4.ptr(), s4.ld, npix16, AF, 1, st));
        } else {
            // D' (denoiser-multi-gpu.py:306-345): dense dilated 3x3 branches, and a real image-level branch:
            // avg-pool 2x2 -> 1x1 conv + bias -> bilinear back to [aspp, aspp] -> BN -> relu6
            conv3x3("aspp_small", cur, s1, curs, ldcs);
            conv3x3("aspp_medium", cur, s2, curs, ldcs);
            conv3x3("aspp_large", cur, s3, curs, ldcs);
            const int Hp = (S16 + 1) / 2;
            T4 pooled = E(B, Hp, Hp, AF);
            if (live()) call(emd_avgpool2x2_f32(cur.ptr(), cur.ld, pooled.ptr(), pooled.ld, B, S16, S16, AF, st));
            T4 img = conv1x1("aspp_image_conv", pooled, nullptr);
            T4 up = E(B, S16, S16, AF);
            if (live()) {
                call(emd_resize_bilinear_f32(img.ptr(), img.ld, up.ptr(), up.ld, B, Hp, Hp, S16, S16, AF, st));
                call(emd_affine_relu6_f32(up.ptr(), up.ld, P["aspp_pooling_bn"].scale, P["aspp_pooling_bn"].shift, s4.ptr(), s4.ld, npix16, AF, 1, st));
            }
            free(pooled); free(img); free(up);
        }
        T4 aspp = conv1x1("aspp_reduce", cat, nullptr);
        free(cur);
        free(cat);
        ar->release(curs);
        // decoder (:350-384)
        T4 c2a = concat2.slice(0, AOUT);
        if (live()) call(emd_resize_bilinear_f32(aspp.ptr(), aspp.ld, c2a.ptr(), c2a.ld, B, S16, S16, S4, S4, AOUT, st));
        free(aspp);
        T4 residual2_d = conv1x1("residual2_d", concat2, nullptr);
        T4 d2a = sep("deconv2_a", concat2, nullptr, nullptr);
        const bool so2 = deconv_fused_ok(B, S4, S4, F2, F2) && S4 % 8 == 0 && S4 % 16 == 0;
        void* deconv2_s = nullptr;
        T4 deconv2 = sep("deconv2_b", d2a, nullptr, &residual2_d, so2 ? &deconv2_s : nullptr);
        free(d2a); free(residual2_d); free(concat2);
        T4 c1a = concat1.slice(0, F2);
        deconv("deconv2to1", so2 ? nullptr : &deconv2, deconv2_s, B, S4, S4, c1a);
        free(deconv2);
        ar->release(deconv2_s);
        // deconv1_a + residual1_d read concat1: one launch (128 | 128 columns) where emd_sep3x3_dual_preferred says so
        T4 residual1_d, d1a;
        sep_and_projection("deconv1_a", "residual1_d", concat1, &d1a, &residual1_d);
        const bool so1 = deconv_fused_ok(B, S2, S2, F1, F1) && S2 % 8 == 0 && S2 % 16 == 0;
        void* deconv1_s = nullptr;
        T4 deconv1 = sep("deconv1_b", d1a, nullptr, &residual1_d, so1 ? &deconv1_s : nullptr);
        free(d1a); free(residual1_d); free(concat1);
        T4 deconv1to0 = E(B, S, S, F1);
        deconv("deconv1to0", so1 ? nullptr : &deconv1, deconv1_s, B, S2, S2, deconv1to0);
        free(deconv1);
        ar->release(deconv1_s);
        T4 d0a, residual0_d;
        sep_and_projection("deconv0_a", "residual0_d", deconv1to0, &d0a, &residual0_d);
        T4 deconv0 = sep("deconv0_b", d0a, nullptr, &residual0_d);
        free(deconv1to0); free(d0a); free(residual0_d);
        if (live()) {
            const LayerParams& pf = P["deconv_final"];   // no output clip in D (:396); D' clips to [0, 1] in-graph (denoiser-multi-gpu.py:534-538)
            call(emd_conv3x3_cout1_f32(deconv0.ptr(), deconv0.ld, pf.wfin, pf.scale_f, pf.shift_f, yout, B, S, S, F0, g->twin ? 2 : 1, 0.f, 0, st));
        }
        free(deconv0);
    }
};

}  // namespace

extern "C" int emd_graph_create(emd_graph** out, int variant, int n_vars, const char* const* names, const float* const* data,
                                const long* counts) {
    EMD_REQUIRE(out && names && data && counts && n_vars > 0, EMD_E_INVALID, "emd_graph_create: null argument");
    EMD_REQUIRE(variant >= 0 && variant <= 3, EMD_E_UNSUPPORTED,
                "emd_graph_create: variant 0 (graph D, machine_learning/denoiser.py), 1 (graph D', misc_py/denoiser-multi-gpu.py, phase=False) , 2 (graph X, misc_py/modified_Xception.py) or 3 (graph G's generator, misc_py/gan-infilling-100.py)");
    *out = nullptr;
    WeightMap w;
    for (int i = 0; i < n_vars; ++i) {
        EMD_REQUIRE(names[i] && data[i] && counts[i] > 0, EMD_E_INVALID, "emd_graph_create: null variable entry");
        w[names[i]] = {data[i], counts[i]};
    }
    emd_graph* g = new emd_graph();
    g->twin = variant == 1;
    std::string err;
    bool ok = true;
    if (variant >= 2) {   // graphs X / G: their own layer tables, parameters and launch sequences (graph_exec_x.hip, graph_exec_g.hip)
        if (variant == 2) g->x = emd::gx::x_create(w, g->allocs, &err);
        else g->gen = emd::gx::g_create(w, g->allocs, &err);
        if (!g->x && !g->gen) {
            const bool dev_failure = err.find("device allocation") != std::string::npos || err.find("upload") != std::string::npos;
            emd::set_error("%s", err.c_str());
            for (void* q : g->allocs) (void)hipFree(q);
            delete g;
            return dev_failure ? EMD_E_ALLOC : EMD_E_INVALID;
        }
        *out = g;
        return EMD_OK;
    }
    const float u4[4] = {1.f, 0.f, 0.f, 0.f}, z4[4] = {0.f, 0.f, 0.f, 0.f};
    g->unit4 = upload(g, u4, 4);
    g->zero4 = upload(g, z4, 4);
    ok = g->unit4 && g->zero4;
    if (!ok) err = "emd_graph_create: device allocation failed";
    for (const LayerDecl& d : declare_layers(g->twin)) {
        if (!ok) break;
        LayerParams p;
        p.d = d;
        std::vector<double> s, t;
        if (d.kind == SEP) {
            const float *dw, *pw;
            ok = fetch(w, d.scope + "/depthwise_weights", 9L * d.cin, &dw, &err) && fetch(w, d.scope + "/pointwise_weights", (long)d.cin * d.cout, &pw, &err) &&
                 fold(w, d, nullptr, &s, &t, &err);
            if (!ok) break;
            if (d.cin == 1) {   // cnn0: depthwise on the 1-channel image, then an outer product
                std::vector<double> a(d.cout);
                for (int c = 0; c < d.cout; ++c) a[c] = (double)pw[c] * (double)(float)s[c];   // the folded scale is a float32 value (as in denoiser.py)
                p.w9 = upload(g, dw, 9);
                p.a = upload_f(g, a);
                p.shift = upload_f(g, t);
                ok = p.w9 && p.a && p.shift;
            } else {
                p.dw = upload(g, dw, 9 * (size_t)d.cin);   // [3][3][Cin][1] == [9][Cin]
                ok = p.dw && pack(g, pw, 1, d.cin, d.cout, 0, &p.pw);
                p.scale = upload_f(g, s);
                p.shift = upload_f(g, t);
                ok = ok && p.scale && p.shift;
            }
            if (ok && !d.extra_bn.empty()) {
                std::vector<double> gs, hs;
                ok = bn_affine(w, d.extra_bn, d.cout, &gs, &hs, &err);
                if (ok) {
                    p.scale2 = upload_f(g, gs);
                    p.shift2 = upload_f(g, hs);
                    ok = p.scale2 && p.shift2;
                }
            }
        } else if (d.kind == CONV) {
            const float *wt, *bias;
            ok = fetch(w, d.scope + "/" + d.wname, (long)d.k * d.k * d.cin * d.cout, &wt, &err) && fetch(w, d.scope + "/" + d.bname, d.cout, &bias, &err) &&
                 fold(w, d, bias, &s, &t, &err);
            if (!ok) break;
            if (d.cin == 1) {   // residual0
                std::vector<double> a(d.cout);
                for (int c = 0; c < d.cout; ++c) a[c] = (double)wt[c] * (double)(float)s[c];
                p.a = upload_f(g, a);
                p.shift = upload_f(g, t);
                ok = p.a && p.shift;
            } else if (d.cout == 1) {   // deconv_final: [3][3][Cin][1] == [9][Cin]
                p.wfin = upload(g, wt, 9 * (size_t)d.cin);
                p.scale_f = (float)s[0];
                p.shift_f = (float)t[0];
                ok = p.wfin != nullptr;
            } else {
                ok = pack(g, wt, d.k * d.k, d.cin, d.cout, 0, &p.pw);
                p.scale = upload_f(g, s);
                p.shift = upload_f(g, t);
                ok = ok && p.scale && p.shift;
            }
        } else if (d.kind == DECONV) {
            const float *wt, *bias;   // [3][3][Cout][Cin]
            ok = fetch(w, d.scope + "/" + d.wname, 9L * d.cin * d.cout, &wt, &err) && fetch(w, d.scope + "/" + d.bname, d.cout, &bias, &err) &&
                 fold(w, d, bias, &s, &t, &err);
            if (!ok) break;
            for (int ph = 0; ph < 4 && ok; ++ph) {
                int ky[4], kx[4];
                const int nt = emd_deconv_phase_taps(ph, ky, kx);
                std::vector<float> sub((size_t)nt * d.cout * d.cin);
                for (int q = 0; q < nt; ++q)
                    std::memcpy(sub.data() + (size_t)q * d.cout * d.cin, wt + (size_t)(ky[q] * 3 + kx[q]) * d.cout * d.cin, sizeof(float) * d.cout * d.cin);
                ok = pack(g, sub.data(), nt, d.cin, d.cout, 1, &p.phase[ph]);
            }
            p.scale = upload_f(g, s);
            p.shift = upload_f(g, t);
            ok = ok && p.scale && p.shift;
        } else {
            std::vector<double> gs, hs;
            ok = bn_affine(w, d.bn[0], d.cout, &gs, &hs, &err);
            if (ok) {
                p.scale = upload_f(g, gs);
                p.shift = upload_f(g, hs);
                ok = p.scale && p.shift;
            }
        }
        if (!ok && err.empty()) err = "emd_graph_create: device allocation or weight packing failed at layer " + d.key;
        g->P[d.key] = p;
    }
    if (!ok) {
        // a missing / mis-sized variable names itself in err; everything else on this path is a device allocation or upload
        const bool dev_failure = err.find("device allocation") != std::string::npos || err.find("upload") != std::string::npos;
        emd::set_error("%s", err.c_str());
        for (void* q : g->allocs) (void)hipFree(q);
        delete g;
        return dev_failure ? EMD_E_ALLOC : EMD_E_INVALID;
    }
    *out = g;
    return EMD_OK;
}

extern "C" size_t emd_graph_workspace_bytes(emd_graph* g, int B, int S) {
    if (!g || B < 1 || S < 16 || S % 16) return 0;
    if (g->x || g->gen) {
        if ((g->x && S % 64) || (g->gen && S < 32)) return 0;
        Arena ar;
        ar.measuring = true;
        const int rc = g->x ? emd::gx::x_forward(g->x, &ar, nullptr, true, nullptr, nullptr, B, S)
                            : emd::gx::g_forward(g->gen, &ar, nullptr, true, nullptr, nullptr, B, S);
        return rc == EMD_OK ? ar.peak + 256 : 0;
    }
    // the larger of the two launch forms: a size asked for before emd_graph_set_two_streams stays valid after it
    size_t need = 0;
    const bool keep = g->two_streams;
    for (int mode = 0; mode < 2; ++mode) {
        g->two_streams = mode != 0;
        Arena ar;
        ar.measuring = true;
        Run r{g, &ar, nullptr, true};
        r.forward(nullptr, nullptr, B, S);
        need = ar.peak > need ? ar.peak : need;
    }
    g->two_streams = keep;
    return need + 256;
}

extern "C" int emd_graph_run(emd_graph* g, const float* x, float* y, int B, int S, void* workspace, size_t workspace_bytes,
                             emd_stream_t stream) {
    EMD_REQUIRE(g && x && y && workspace, EMD_E_INVALID, "emd_graph_run: null pointer");
    EMD_REQUIRE(B >= 1 && S >= 16 && S % 16 == 0, EMD_E_INVALID, "emd_graph_run: square crops with side a multiple of 16, B >= 1");
    EMD_REQUIRE(x != y, EMD_E_INVALID, "emd_graph_run: the output aliases the input");
    unsigned char* base = static_cast<unsigned char*>(workspace);
    const size_t skew = (256 - (reinterpret_cast<uintptr_t>(base) & 255)) & 255;
    EMD_REQUIRE(workspace_bytes > skew, EMD_E_INVALID, "emd_graph_run: workspace too small");
    Arena ar;
    ar.base = base + skew;
    ar.cap = workspace_bytes - skew;
    if (g->x) {
        EMD_REQUIRE(S % 64 == 0, EMD_E_INVALID, "emd_graph_run: graph X takes square crops with side a multiple of 64");
        return emd::gx::x_forward(g->x, &ar, static_cast<hipStream_t>(stream), false, x, y, B, S);
    }
    if (g->gen) {
        EMD_REQUIRE(S >= 32, EMD_E_INVALID, "emd_graph_run: graph G takes square crops with side a multiple of 16, >= 32");
        return emd::gx::g_forward(g->gen, &ar, static_cast<hipStream_t>(stream), false, x, y, B, S);
    }
    Run r{g, &ar, static_cast<hipStream_t>(stream), false};
    r.forward(x, y, B, S);
    return r.rc;
}

extern "C" int emd_graph_set_two_streams(emd_graph* g, int on) {
    EMD_REQUIRE(g, EMD_E_INVALID, "emd_graph_set_two_streams: null handle");
    g->two_streams = on != 0;
    return EMD_OK;
}

extern "C" void emd_graph_destroy(emd_graph* g) {
    if (!g) return;
    for (void* q : g->allocs) (void)hipFree(q);
    g->drop_streams();
    if (g->x) emd::gx::x_destroy(g->x);
    if (g->gen) emd::gx::g_destroy(g->gen);
    delete g;
}
