// Library-level entry points of the C ABI (include/street_crafter_amd.h).
#include "raster_common.h"
#include <string.h>

extern "C" const char* sc_version(void) { return "street_crafter_amd 0.1.0 (gfx950)"; }

extern "C" const char* sc_target_arch(void) { return "gfx950"; }

extern "C" const char* sc_error_string(int code) {
    switch (code) {
        case SC_OK: return "ok";
        case SC_EINVAL: return "street_crafter_amd: invalid argument (size / null pointer / unsupported parameter)";
        case SC_EWORKSPACE: return "street_crafter_amd: workspace too small";
        case SC_EUNSUPPORTED: return "street_crafter_amd: unsupported configuration";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "street_crafter_amd: unknown error";
}

int g_sc_debug[4] = {0, 0, 0, 0};

extern "C" int sc_set_option(const char* key, int value) {
    if (!key) return SC_EINVAL;
    if (strncmp(key, "debug", 5) == 0 && key[5] >= '0' && key[5] <= '3' && key[6] == 0) {
        if (value < 0) return SC_EINVAL;
        const int prev = g_sc_debug[key[5] - '0'];
        g_sc_debug[key[5] - '0'] = value;
        return prev;
    }
    if (strcmp(key, "raster_fwd") == 0) {
        if (value < 0 || value > 1) return SC_EINVAL;
        const int prev = g_sc_raster_fwd_variant;
        g_sc_raster_fwd_variant = value;
        return prev;
    }
    return SC_EINVAL;
}
