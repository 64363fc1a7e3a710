// the caller-side glue of street_gaussian/models/street_gaussian_renderer.py folded into
// the kernels, behind gsplat's own one-call API `rasterization()` (imported at renderer.py:204):
//
//   projection_sh_fwd_kernel = a1 fully_fused_projection                       (renderer.py:219-234)
//                            + a2 opacities * compensations                      (:235-238)
//                            + a5 dirs = xyz - camera_center, masks = radii > 0   (:256-258)
//                            + a6 spherical_harmonics                            (:259)
//                            + a7 clamp_min(colors + 0.5, 0), cat(colors, depth)  (:260, :265-266)
//
// one lane per (camera, Gaussian): 92 B in (mean 12, quat 16, scale 12, opacity 4, SH 12 K), 48 B out
// (radius 4, mean2d 8, depth 4, conic 12, opacity 4, colour+depth 16) instead of the ~245 B the
// separate operators and the torch elementwise kernels between them move.  Every value is computed by
// the SAME device functions in the SAME order as the separate operators (projection_common.h,
// sh_common.h, plain IEEE mul / sub / add for the glue), so the fused path is bit-identical to the
// composed one; tests/test_gpu_parity.py::test_rasterization_fused_matches_composition checks that.
#include "projection_common.h"
#include "sh_common.h"

#pragma clang fp contract(off)

namespace {

// camera centre = -R^T t of a rigid world-to-camera matrix (row-major [R|t]); one lane per camera
__global__ void camera_centers_kernel(const float* __restrict__ viewmats, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* V = viewmats + (size_t)c * 16;
    const float tx = V[3], ty = V[7], tz = V[11];
    out[c * 3 + 0] = -dot3(V[0], tx, V[4], ty, V[8], tz);
    out[c * 3 + 1] = -dot3(V[1], tx, V[5], ty, V[9], tz);
    out[c * 3 + 2] = -dot3(V[2], tx, V[6], ty, V[10], tz);
}

template <int DEG>
__global__ __launch_bounds__(256) void projection_sh_fwd_kernel(
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ opacities, const float* __restrict__ coeffs,
    const float* __restrict__ viewmats, const float* __restrict__ Ks, const float* __restrict__ campos,
    int N, int K, int width, int height, float eps2d, float near_plane, float far_plane,
    float radius_clip, ProjOpt opt, int antialiased, int32_t* __restrict__ radii, float* __restrict__ means2d,
    float* __restrict__ depths, float* __restrict__ conics, float* __restrict__ opac_out,
    float* __restrict__ colors4, float4* __restrict__ records) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int cam = blockIdx.y;
    if (n >= N) return;
    const Cam c = load_cam(viewmats + cam * 16, Ks + cam * 9);
    const size_t o = (size_t)cam * N + n;
    const ProjOut p = project_one(c, means, quats, scales, n, width, height, eps2d, near_plane, far_plane,
                                  radius_clip, opt);
    radii[o] = p.rad_i;
    *reinterpret_cast<float2*>(means2d + o * 2) = make_float2(p.m2x, p.m2y);
    depths[o] = p.depth;
    // (with `records`, conics / opacities / colours may be left out: sc_records_unpack rebuilds them on demand)
    if (conics) {
        conics[o * 3 + 0] = p.con0;
        conics[o * 3 + 1] = p.con1;
        conics[o * 3 + 2] = p.con2;
    }
    // a2: culled rows carry compensation 0, hence opacity 0, exactly as the torch product does
    const float op = opacities[n];
    const float opo = antialiased ? op * p.comp : op;
    if (opac_out) opac_out[o] = opo;
    // a5 + a6: masked-out rows evaluate to 0 (sh_fwd_kernel), then a7 turns that into 0.5
    float r = 0.f, g = 0.f, b = 0.f;
    if (p.rad_i > 0) {
        const float dx = means[n * 3 + 0] - campos[cam * 3 + 0];
        const float dy = means[n * 3 + 1] - campos[cam * 3 + 1];
        const float dz = means[n * 3 + 2] - campos[cam * 3 + 2];
        sh_eval<DEG>(dx, dy, dz, coeffs + (size_t)n * K * 3, r, g, b);
    }
    // a7: clamp_min(x + 0.5, 0) keeps NaN like torch does
    r = r + 0.5f; g = g + 0.5f; b = b + 0.5f;
    r = (r < 0.f) ? 0.f : r; g = (g < 0.f) ? 0.f : g; b = (b < 0.f) ? 0.f : b;
    if (colors4) *reinterpret_cast<float4*>(colors4 + o * 4) = make_float4(r, g, b, p.depth);
    // the rasterizer's record (sc_rasterize_fwd_packed): everything it gathers per splat in one 48-B line
    if (records) {
        records[o * 3 + 0] = make_float4(p.m2x, p.m2y, p.con0, p.con1);
        records[o * 3 + 1] = make_float4(p.con2, opo, r, g);
        records[o * 3 + 2] = make_float4(b, p.depth, 0.f, 0.f);
    }
}

// conics / opacities / colours of `meta` from the rasterizer's records (first access only: the fused frame itself
// never reads them)
__global__ __launch_bounds__(256) void records_unpack_kernel(const float4* __restrict__ records, int64_t CN,
                                                             float* __restrict__ conics, float* __restrict__ opac,
                                                             float* __restrict__ colors4) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= CN) return;
    const float4 q0 = records[o * 3 + 0], q1 = records[o * 3 + 1], q2 = records[o * 3 + 2];
    if (conics) { conics[o * 3 + 0] = q0.z; conics[o * 3 + 1] = q0.w; conics[o * 3 + 2] = q1.x; }
    if (opac) opac[o] = q1.y;
    if (colors4) *reinterpret_cast<float4*>(colors4 + o * 4) = make_float4(q1.z, q1.w, q2.x, q2.y);
}

}  // namespace

#pragma clang fp contract(fast)

extern "C" int sc_camera_centers(const float* viewmats, int C, float* out, sc_stream_t stream) {
    if (C < 0) return SC_EINVAL;
    if (C == 0) return SC_OK;
    if (!viewmats || !out) return SC_EINVAL;
    hipLaunchKernelGGL(camera_centers_kernel, dim3((C + 63) / 64), dim3(64), 0, sc_s(stream), viewmats, C, out);
    SC_LAUNCH_CHECK();
    return SC_OK;
}

extern "C" int sc_projection_sh_fwd(const float* means, const float* quats, const float* scales,
                                    const float* opacities, const float* sh_coeffs, const float* viewmats,
                                    const float* Ks, const float* camera_centers, int C, int N, int K,
                                    int sh_degree, int width, int height, float eps2d, float near_plane,
                                    float far_plane, float radius_clip, int antialiased, int32_t* radii,
                                    float* means2d, float* depths, float* conics, float* opacities_out,
                                    float* colors4, float* records, sc_stream_t stream) {
    if (C < 0 || N < 0 || width <= 0 || height <= 0) return SC_EINVAL;
    if (sh_degree < 0 || sh_degree > 4 || K < (sh_degree + 1) * (sh_degree + 1)) return SC_EINVAL;
    if (C == 0 || N == 0) return SC_OK;
    if (!means || !quats || !scales || !opacities || !sh_coeffs || !viewmats || !Ks || !camera_centers ||
        !radii || !means2d || !depths || (!records && (!conics || !opacities_out || !colors4)))
        return SC_EINVAL;
    if (records && ((uintptr_t)records & 15)) return SC_EINVAL;
    if (C > 65535) return SC_EINVAL;
    dim3 grid((N + 255) / 256, C);
#define SC_LAUNCH_FUSED(DEG)                                                                                 \
    hipLaunchKernelGGL(projection_sh_fwd_kernel<DEG>, grid, dim3(256), 0, sc_s(stream), means, quats, scales,  \
                       opacities, sh_coeffs, viewmats, Ks, camera_centers, N, K, width, height, eps2d,        \
                       near_plane, far_plane, radius_clip, sc_proj_opt(), antialiased, radii, means2d, depths, conics, \
                       opacities_out, colors4, reinterpret_cast<float4*>(records))
    switch (sh_degree) {
        case 0: SC_LAUNCH_FUSED(0); break;
        case 1: SC_LAUNCH_FUSED(1); break;
        case 2: SC_LAUNCH_FUSED(2); break;
        case 3: SC_LAUNCH_FUSED(3); break;
        default: SC_LAUNCH_FUSED(4); break;
    }
#undef SC_LAUNCH_FUSED
    SC_LAUNCH_CHECK();
    return SC_OK;
}


extern "C" int sc_records_unpack(const float* records, int64_t CN, float* conics, float* opacities, float* colors4,
                                 sc_stream_t stream) {
    if (CN < 0) return SC_EINVAL;
    if (CN == 0) return SC_OK;
    if (!records || ((uintptr_t)records & 15)) return SC_EINVAL;
    const int64_t nb = (CN + 255) / 256;
    if (nb > 0x7fffffff) return SC_EINVAL;
    hipLaunchKernelGGL(records_unpack_kernel, dim3((unsigned)nb), dim3(256), 0, sc_s(stream),
                       reinterpret_cast<const float4*>(records), CN, conics, opacities, colors4);
    SC_LAUNCH_CHECK();
    return SC_OK;
}
