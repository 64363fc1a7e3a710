// Fragment layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, found by experiment: lane l supplies a = 1000 + l, b = the l-th prime-ish
// tag; D[v] of every lane is printed as the (a-lane, b-lane) pair that produced it.  Also: is D = fmaf(a, b, C) bit for bit?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, const float* a_in, const float* b_in, const float* c_in) {
    const int l = threadIdx.x;
    f4 c = {c_in[l * 4 + 0], c_in[l * 4 + 1], c_in[l * 4 + 2], c_in[l * 4 + 3]};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a_in[l], b_in[l], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
    float a[64], b[64], c[256], o[256], *da, *db, *dc, *d_o;
    hipMalloc(&da, sizeof a); hipMalloc(&db, sizeof b); hipMalloc(&dc, sizeof c); hipMalloc(&d_o, sizeof o);
    // layout: a = 2^l-free encoding: a_l = 1 + l, b_l = 128 * (1 + l): product = 128 (1 + la)(1 + lb) identifies the pair
    for (int l = 0; l < 64; ++l) { a[l] = 1.f + l; b[l] = 128.f * (1.f + l); }
    memset(c, 0, sizeof c);
    hipMemcpy(da, a, sizeof a, hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof b, hipMemcpyHostToDevice); hipMemcpy(dc, c, sizeof c, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_o, da, db, dc);
    hipMemcpy(o, d_o, sizeof o, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            const int la = (l / 4) * 4 + v, lb = l;             // the guess: D[v] of lane (block, j) = a(lane (block, v)) * b(own lane)
            const float want = a[la] * b[lb];
            if (o[l * 4 + v] != want) { ok = 0; if (l < 8) printf("lane %d v %d: got %g, guess %g\n", l, v, o[l * 4 + v], want); }
        }
    printf("layout guess D[v](lane 4q + j) = a(lane 4q + v) * b(lane 4q + j): %s\n", ok ? "CONFIRMED" : "WRONG");
    if (!ok) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    // numerics: D == fmaf(a, b, C) bit for bit, on awkward values (incl. subnormal products and sums)
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed; };
    int bad = 0, total = 0;
    for (int rep = 0; rep < 200; ++rep) {
        for (int l = 0; l < 64; ++l) {
            auto f = [&](int mode) { unsigned u = rnd(); float x; if (mode == 0) { u = (u & 0x807fffffu) | ((100u + (rnd() % 60u)) << 23); } else if (mode == 1) { u = (u & 0x807fffffu) | ((rnd() % 30u) << 23); } memcpy(&x, &u, 4); return x; };
            const int mode = rep % 3 == 2 ? 1 : 0;
            a[l] = f(mode); b[l] = f(0);
            for (int v = 0; v < 4; ++v) c[l * 4 + v] = f(mode);
        }
        hipMemcpy(da, a, sizeof a, hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof b, hipMemcpyHostToDevice); hipMemcpy(dc, c, sizeof c, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d_o, da, db, dc);
        hipMemcpy(o, d_o, sizeof o, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int v = 0; v < 4; ++v) {
                const float want = fmaf(a[(l / 4) * 4 + v], b[l], c[l * 4 + v]);
                ++total;
                if (memcmp(&want, &o[l * 4 + v], 4) != 0 && !(std::isnan(want) && std::isnan(o[l * 4 + v]))) {
                    if (bad < 5) printf("mismatch: a %a b %a c %a: mfma %a fmaf %a\n", a[(l / 4) * 4 + v], b[l], c[l * 4 + v], o[l * 4 + v], want);
                    ++bad;
                }
            }
    }
    printf("D == fmaf(a, b, C) bit for bit: %d of %d differ\n", bad, total);
    return 0;
}
