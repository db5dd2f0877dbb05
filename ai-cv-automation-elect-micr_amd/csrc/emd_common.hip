// emd_version / emd_last_error and the thread-local error buffer.
#include <cstring>

#include "emd_common.hpp"

namespace emd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

Knobs g_knobs;

}  // namespace emd

extern "C" int emd_version(void) { return EMD_VERSION; }

extern "C" const char* emd_last_error(void) { return emd::g_err; }

// dev hook (include/emdenoise_dev.h): sets one of emd::Knobs by name; returns 0, or -1 for an unknown name
extern "C" int emd_debug_knob(const char* name, long value) {
    using emd::g_knobs;
    struct { const char* n; int* v; } tab[] = {
        {"sep_pipe", &g_knobs.sep_pipe}, {"sep_pipe2", &g_knobs.sep_pipe2}, {"sep_stamp_wave", &g_knobs.sep_stamp_wave}, {"sep_mode", &g_knobs.sep_mode}, {"sep_tpw", &g_knobs.sep_tpw}, {"sep_ablate", &g_knobs.sep_ablate}, {"sep_nw", &g_knobs.sep_nw}, {"sep_xcd", &g_knobs.sep_xcd},
        {"sep_wide", &g_knobs.sep_wide}, {"sep_wres", &g_knobs.sep_wres}, {"nt_mask", &g_knobs.nt_mask}, {"deconv_direct", &g_knobs.deconv_direct}, {"epi_width", &g_knobs.epi_width}, {"split_lead", &g_knobs.split_lead}, {"dw_xcd", &g_knobs.dw_xcd},
        {"dw_th", &g_knobs.dw_th},       {"split_variant", &g_knobs.split_variant}, {"split_narrow", &g_knobs.split_narrow}, {"split_wide", &g_knobs.split_wide}, {"conv3_pipe", &g_knobs.conv3_pipe}, {"wgrad_msplit", &g_knobs.wgrad_msplit}, {"wgrad_tile", &g_knobs.wgrad_tile}, {"sep_gen_pipe", &g_knobs.sep_gen_pipe},
    };
    if (!name) return -1;
    for (auto& e : tab)
        if (!strcmp(name, e.n)) { *e.v = (int)value; return 0; }
    return -1;
}
