// Shared pieces of the native graph executors (graph_exec.hip: graphs D / D'; graph_exec_x.hip: graph X): host-side weight lookup,
// upload and packing, the workspace arena, activation views.
#pragma once

#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "mfma_common.hpp"

namespace emd {
namespace gx {

struct Packed {   // bf16 hi / lo planes on the device (one allocation, lo behind hi)
    uint16_t* hi = nullptr;
    uint16_t* lo = nullptr;
};

typedef std::map<std::string, std::pair<const float*, long>> WeightMap;

inline bool fetch(const WeightMap& w, const std::string& name, long count, const float** out, std::string* err) {
    auto it = w.find(name);
    if (it == w.end()) {
        *err = "emd_graph_create: missing variable " + name;
        return false;
    }
    if (it->second.second != count) {
        *err = "emd_graph_create: " + name + ": " + std::to_string(it->second.second) + " elements, expected " + std::to_string(count);
        return false;
    }
    *out = it->second.first;
    return true;
}

template <typename T>
T* upload(std::vector<void*>& allocs, const T* host, size_t n) {
    void* d = nullptr;
    if (hipMalloc(&d, n * sizeof(T) < 16 ? 16 : n * sizeof(T)) != hipSuccess) return nullptr;
    allocs.push_back(d);
    if (hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return static_cast<T*>(d);
}

inline float* upload_f(std::vector<void*>& allocs, const std::vector<double>& v) {
    std::vector<float> f(v.begin(), v.end());
    return upload(allocs, f.data(), f.size());
}

// host weights [taps][a][b] -> packed planes on the device
inline bool pack(std::vector<void*>& allocs, const float* w, int taps, int cin, int cout, int cout_major, Packed* out) {
    const size_t n = emd_packed_weight_elems(taps, cin, cout), npad = (n + 63) / 64 * 64;
    std::vector<uint16_t> both(2 * npad, 0);
    if (emd_pack_weights_bf16(w, taps, cin, cout, cout_major, both.data(), both.data() + npad) != EMD_OK) return false;
    uint16_t* d = upload(allocs, both.data(), both.size());
    if (!d) return false;
    out->hi = d;
    out->lo = d + npad;
    return true;
}

// inference batch norm as y = x * gs + hs (float64)
inline bool bn_affine(const WeightMap& w, const std::string& scope, int C, double eps, std::vector<double>* gs, std::vector<double>* hs,
                      std::string* err) {
    const float *gamma, *beta, *mean, *var;
    if (!fetch(w, scope + "/gamma", C, &gamma, err) || !fetch(w, scope + "/beta", C, &beta, err) ||
        !fetch(w, scope + "/moving_mean", C, &mean, err) || !fetch(w, scope + "/moving_variance", C, &var, err))
        return false;
    gs->resize(C);
    hs->resize(C);
    for (int c = 0; c < C; ++c) {
        (*gs)[c] = (double)gamma[c] / std::sqrt((double)var[c] + eps);
        (*hs)[c] = (double)beta[c] - (double)mean[c] * (*gs)[c];
    }
    return true;
}

// ---- workspace: a first-fit free-list allocator over the caller's buffer; in measuring mode it only tracks the peak
struct Arena {
    unsigned char* base = nullptr;
    size_t cap = 0, peak = 0;
    bool measuring = false;
    std::map<size_t, size_t> live;   // offset -> size
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        size_t off = 0;
        for (auto& kv : live) {   // ordered by offset: first gap that fits
            if (kv.first - off >= bytes) break;
            off = kv.first + kv.second;
        }
        if (!measuring && off + bytes > cap) return nullptr;
        live[off] = bytes;
        if (off + bytes > peak) peak = off + bytes;
        return measuring ? reinterpret_cast<void*>(off + 4096) : static_cast<void*>(base + off);   // measuring: a fake non-null address
    }
    void release(void* p) {
        if (!p) return;
        const size_t off = measuring ? reinterpret_cast<size_t>(p) - 4096 : static_cast<size_t>(static_cast<unsigned char*>(p) - base);
        live.erase(off);
    }
};

struct T4 {   // an activation: channels [c0, c0 + C) of a [B,H,W,ld] fp32 buffer
    float* buf = nullptr;
    int B = 0, H = 0, W = 0, C = 0, ld = 0, c0 = 0;
    float* ptr() const { return buf + c0; }
    T4 slice(int off, int n) const {
        T4 t = *this;
        t.c0 = c0 + off;
        t.C = n;
        return t;
    }
};

// ---- graph X (graph_exec_x.hip): misc_py/modified_Xception.py:194-654, inference
struct XGraph;
XGraph* x_create(const WeightMap& w, std::vector<void*>& allocs, std::string* err);
void x_destroy(XGraph* x);
// launches (dry = false) or only plans the workspace (dry = true: ar in measuring mode); returns an EMD_* code
int x_forward(XGraph* x, Arena* ar, hipStream_t st, bool dry, const float* in, float* out, int B, int S);

// ---- graph G's generator (graph_exec_g.hip): misc_py/gan-infilling-100.py:133-374, inference
struct GGraph;
GGraph* g_create(const WeightMap& w, std::vector<void*>& allocs, std::string* err);
void g_destroy(GGraph* g);
int g_forward(GGraph* g, Arena* ar, hipStream_t st, bool dry, const float* in, float* out, int B, int S);

}  // namespace gx
}  // namespace emd
