// Tile-bucketed intersection path (placeholder: implemented in the next milestone).
#include "sc_common.h"
extern "C" size_t sc_isect_bin_workspace_bytes(int64_t, int, int, int, int64_t) { return 256; }
extern "C" int sc_isect_bin_count(const float*, const int32_t*, int, int, int, int, int, int32_t*, int32_t*,
                                  int64_t*, void*, size_t, sc_stream_t) { return SC_EUNSUPPORTED; }
extern "C" int sc_isect_bin_sort(const float*, const int32_t*, const float*, int, int, int, int, int,
                                 const int32_t*, int64_t, int64_t*, int32_t*, void*, size_t, sc_stream_t) {
    return SC_EUNSUPPORTED;
}
