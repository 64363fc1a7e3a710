// Host-side binding layer: Python <-> the C ABI of include/street_crafter_amd.h, compiled (g++, no device code).
//
// WHY.  The operators keep the reference's Python signatures (gsplat.rendering.*, called one by one from
// street_gaussian/models/street_gaussian_renderer.py:219-280), so every frame crosses Python -> native five times.
// Through ctypes a crossing costs ~6-8 us of argument conversion, and every output / workspace tensor a separate
// ~3-4 us torch.empty from Python: ~150 us of host time per frame (tools/exp_host.py), which is what bounds small
// scenes (S-100k: 0.24-0.30 ms per frame, all of it host) and makes the training step's rate depend on how busy the
// box's host is (BENCH_r02: 786 steps/s on the driver's box, 1060 on a quiet one, for 0.78 ms of kernels).
// Here one call per operator validates its tensors, allocates ALL of its outputs and scratch through torch's
// caching allocator (at::empty: stream-ordered reuse, no hipMalloc) and calls the SAME C-ABI entry point.  Nothing
// else moves: the C ABI stays the boundary, kernels and results are untouched, autograd wiring stays in
// rendering.py.  torch is used for what it is here for: device memory.
//
// The library's entry points are resolved at link time (libstreet_crafter_hip.so, rpath $ORIGIN).  Streams arrive as
// the raw hipStream_t value (torch._C._cuda_getCurrentRawStream), so no HIP header is needed.
#include <torch/extension.h>

#include <cstdint>
#include <tuple>
#include <vector>

#include "../../include/street_crafter_amd.h"

#ifndef SC_ABI_HASH
#define SC_ABI_HASH "unhashed"
#endif

namespace {

using at::Tensor;
using OptT = c10::optional<Tensor>;

inline sc_stream_t S(int64_t s) { return reinterpret_cast<sc_stream_t>(static_cast<uintptr_t>(s)); }

inline void req(const Tensor& t, at::ScalarType dt, const char* name) {
    TORCH_CHECK(t.is_cuda(), name, " must live on a HIP device (got ", t.device(), "); street_crafter_amd has no CPU path");
    TORCH_CHECK(t.scalar_type() == dt, name, " must be ", dt, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
inline const float* fp(const Tensor& t) { return static_cast<const float*>(t.data_ptr()); }
inline float* fpw(const Tensor& t) { return static_cast<float*>(t.data_ptr()); }
inline const float* fpo(const OptT& t) { return t.has_value() ? static_cast<const float*>(t->data_ptr()) : nullptr; }
inline const int32_t* ip(const Tensor& t) { return static_cast<const int32_t*>(t.data_ptr()); }
inline at::TensorOptions f32(const Tensor& like) { return like.options().dtype(at::kFloat); }
inline at::TensorOptions i32(const Tensor& like) { return like.options().dtype(at::kInt); }
inline at::TensorOptions u8(const Tensor& like) { return like.options().dtype(at::kByte); }

// ---- a1 ---------------------------------------------------------------------------------------------------
// -> (rc, radii, means2d, depths, conics, compensations | None)
struct ProjOutT { int rc; Tensor radii, means2d, depths, conics; OptT comps; };
ProjOutT projection_fwd_c(const Tensor& means, const Tensor& quats, const Tensor& scales, const Tensor& viewmats,
                          const Tensor& Ks, int64_t width, int64_t height, double eps2d, double near_plane,
                          double far_plane, double radius_clip, bool calc_comp, int64_t stream) {
    req(means, at::kFloat, "means"); req(quats, at::kFloat, "quats"); req(scales, at::kFloat, "scales");
    req(viewmats, at::kFloat, "viewmats"); req(Ks, at::kFloat, "Ks");
    const int64_t C = viewmats.size(0), N = means.size(0);
    Tensor radii = at::empty({C, N}, i32(means));
    Tensor means2d = at::empty({C, N, 2}, f32(means));
    Tensor depths = at::empty({C, N}, f32(means));
    Tensor conics = at::empty({C, N, 3}, f32(means));
    OptT comps;
    if (calc_comp) comps = at::empty({C, N}, f32(means));
    const int rc = sc_projection_fwd(fp(means), fp(quats), fp(scales), fp(viewmats), fp(Ks), (int)C, (int)N, (int)width,
                                     (int)height, (float)eps2d, (float)near_plane, (float)far_plane, (float)radius_clip,
                                     static_cast<int32_t*>(radii.data_ptr()), fpw(means2d), fpw(depths), fpw(conics),
                                     comps ? fpw(*comps) : nullptr, S(stream));
    return {rc, radii, means2d, depths, conics, comps};
}
py::tuple projection_fwd(const Tensor& means, const Tensor& quats, const Tensor& scales, const Tensor& viewmats,
                         const Tensor& Ks, int64_t width, int64_t height, double eps2d, double near_plane,
                         double far_plane, double radius_clip, bool calc_comp, int64_t stream) {
    ProjOutT o = projection_fwd_c(means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                                  radius_clip, calc_comp, stream);
    return py::make_tuple(o.rc, o.radii, o.means2d, o.depths, o.conics, o.comps);
}

// -> (rc, v_means, v_quats, v_scales)
struct ProjBwdT { int rc; Tensor v_means, v_quats, v_scales; };
ProjBwdT projection_bwd_c(const Tensor& means, const Tensor& quats, const Tensor& scales, const Tensor& viewmats,
                          const Tensor& Ks, int64_t width, int64_t height, double eps2d, const Tensor& radii,
                          const Tensor& conics, const OptT& comps, const Tensor& v_means2d, const Tensor& v_depths,
                          const Tensor& v_conics, const OptT& v_comps, int64_t stream) {
    req(v_means2d, at::kFloat, "v_means2d"); req(v_depths, at::kFloat, "v_depths"); req(v_conics, at::kFloat, "v_conics");
    if (v_comps) req(*v_comps, at::kFloat, "v_compensations");
    const int64_t C = viewmats.size(0), N = means.size(0);
    Tensor v_means = at::empty_like(means), v_quats = at::empty_like(quats), v_scales = at::empty_like(scales);
    const int rc = sc_projection_bwd(fp(means), fp(quats), fp(scales), fp(viewmats), fp(Ks), (int)C, (int)N, (int)width,
                                     (int)height, (float)eps2d, ip(radii), fp(conics), fpo(comps), fp(v_means2d),
                                     fp(v_depths), fp(v_conics), fpo(v_comps), fpw(v_means), fpw(v_quats),
                                     fpw(v_scales), S(stream));
    return {rc, v_means, v_quats, v_scales};
}
py::tuple projection_bwd(const Tensor& means, const Tensor& quats, const Tensor& scales, const Tensor& viewmats,
                         const Tensor& Ks, int64_t width, int64_t height, double eps2d, const Tensor& radii,
                         const Tensor& conics, const OptT& comps, const Tensor& v_means2d, const Tensor& v_depths,
                         const Tensor& v_conics, const OptT& v_comps, int64_t stream) {
    ProjBwdT o = projection_bwd_c(means, quats, scales, viewmats, Ks, width, height, eps2d, radii, conics, comps, v_means2d,
                                  v_depths, v_conics, v_comps, stream);
    return py::make_tuple(o.rc, o.v_means, o.v_quats, o.v_scales);
}

// ---- a3 (tile-bucketed route) -------------------------------------------------------------------------------
// count phase: allocates tiles_per_gauss, isect_offsets, meta_dev, the count workspace and (want_order) the dispatch
// list.  -> (rc, tiles_per_gauss, offsets, meta_dev, count_ws, tile_order | None)
py::tuple isect_bin_count(const Tensor& means2d, const Tensor& radii, const Tensor& depths, int64_t tile_size,
                          int64_t tile_width, int64_t tile_height, const OptT& tile_work, const OptT& viewmats,
                          const OptT& registry, bool want_order, int64_t meta_host_ptr, int64_t seq, int64_t stream) {
    req(means2d, at::kFloat, "means2d"); req(radii, at::kInt, "radii"); req(depths, at::kFloat, "depths");
    const int64_t C = radii.size(0), N = radii.size(1);
    Tensor tpg = at::empty({C, N}, i32(means2d));
    Tensor offsets = at::empty({C, tile_height, tile_width}, i32(means2d));
    Tensor meta_dev = at::empty({4}, means2d.options().dtype(at::kLong));
    size_t wsb = sc_isect_bin_workspace_bytes(C * N, (int)C, (int)tile_width, (int)tile_height, -1);
    if (wsb < 256) wsb = 256;
    Tensor ws0 = at::empty({(int64_t)wsb}, u8(means2d));
    OptT order;
    if (want_order) order = at::empty({(int64_t)sc_tile_order_len((int)(C * tile_width * tile_height))}, i32(means2d));
    const int rc = sc_isect_bin_count(
        fp(means2d), ip(radii), fp(depths), (int)C, (int)N, (int)tile_size, (int)tile_width, (int)tile_height,
        static_cast<int32_t*>(tpg.data_ptr()), static_cast<int32_t*>(offsets.data_ptr()),
        static_cast<int64_t*>(meta_dev.data_ptr()), reinterpret_cast<int64_t*>(static_cast<uintptr_t>(meta_host_ptr)), seq,
        ws0.data_ptr(), wsb, (want_order && tile_work) ? ip(*tile_work) : nullptr, viewmats ? fp(*viewmats) : nullptr,
        registry ? static_cast<int32_t*>(registry->data_ptr()) : nullptr,
        order ? static_cast<int32_t*>(order->data_ptr()) : nullptr, S(stream));
    return py::make_tuple(rc, tpg, offsets, meta_dev, ws0, order);
}

// sort phase: allocates flatten_ids (+ isect_ids when want_ids) for `capacity` elements and the sort workspace (freed
// on return: the caching allocator reuses a block only for LATER work of the same stream).
// -> (rc, isect_ids | None, flatten_ids)
py::tuple isect_bin_sort(const Tensor& means2d, const Tensor& radii, const Tensor& depths, int64_t tile_size,
                         int64_t tile_width, int64_t tile_height, const Tensor& offsets, const Tensor& meta_dev,
                         const Tensor& ws0, int64_t capacity, int64_t rec_capacity, int64_t super_capacity, bool want_ids,
                         int64_t stream) {
    const int64_t C = radii.size(0), N = radii.size(1);
    OptT ids;
    if (want_ids) ids = at::empty({capacity}, means2d.options().dtype(at::kLong));
    Tensor fids = at::empty({capacity}, i32(means2d));
    size_t wsb = sc_isect_bin_workspace_bytes(C * N, (int)C, (int)tile_width, (int)tile_height, rec_capacity);
    if (wsb < 256) wsb = 256;
    Tensor ws = at::empty({(int64_t)wsb}, u8(means2d));
    const int rc = sc_isect_bin_sort(fp(means2d), ip(radii), fp(depths), (int)C, (int)N, (int)tile_size, (int)tile_width,
                                     (int)tile_height, ip(offsets), static_cast<const int64_t*>(meta_dev.data_ptr()),
                                     ws0.data_ptr(), capacity, rec_capacity, super_capacity,
                                     ids ? static_cast<int64_t*>(ids->data_ptr()) : nullptr,
                                     static_cast<int32_t*>(fids.data_ptr()), ws.data_ptr(), wsb, S(stream));
    return py::make_tuple(rc, ids, fids);
}

// host-side wait for the counts (the GIL is released: another host thread keeps launching)
int wait_i64(int64_t addr, int64_t value, int64_t timeout_us) {
    py::gil_scoped_release nogil;
    return sc_wait_i64(reinterpret_cast<const int64_t*>(static_cast<uintptr_t>(addr)), value, timeout_us);
}

// ---- a6 ---------------------------------------------------------------------------------------------------
struct ShFwdT { int rc; Tensor colors; };
ShFwdT sh_fwd_c(int64_t degree, const Tensor& dirs, const Tensor& coeffs, const OptT& masks, int64_t stream) {
    req(dirs, at::kFloat, "dirs"); req(coeffs, at::kFloat, "coeffs");
    if (masks) req(*masks, at::kByte, "masks");
    const int64_t M = dirs.numel() / 3, K = coeffs.size(-2);
    Tensor colors = at::empty(dirs.sizes(), f32(dirs));
    const int rc = sc_sh_fwd((int)degree, fp(dirs), fp(coeffs), masks ? static_cast<const uint8_t*>(masks->data_ptr()) : nullptr,
                             M, (int)K, fpw(colors), S(stream));
    return {rc, colors};
}
py::tuple sh_fwd(int64_t degree, const Tensor& dirs, const Tensor& coeffs, const OptT& masks, int64_t stream) {
    ShFwdT o = sh_fwd_c(degree, dirs, coeffs, masks, stream);
    return py::make_tuple(o.rc, o.colors);
}

struct ShBwdT { int rc; Tensor v_coeffs; OptT v_dirs; };
ShBwdT sh_bwd_c(int64_t degree, const Tensor& dirs, const Tensor& coeffs, const OptT& masks, const Tensor& v_colors,
                bool need_dirs, int64_t stream) {
    req(v_colors, at::kFloat, "v_colors");
    const int64_t M = dirs.numel() / 3, K = coeffs.size(-2);
    Tensor v_coeffs = at::empty_like(coeffs);
    OptT v_dirs;
    if (need_dirs) v_dirs = at::empty_like(dirs);
    const int rc = sc_sh_bwd((int)degree, fp(dirs), fp(coeffs), masks ? static_cast<const uint8_t*>(masks->data_ptr()) : nullptr,
                             M, (int)K, fp(v_colors), fpw(v_coeffs), v_dirs ? fpw(*v_dirs) : nullptr, S(stream));
    return {rc, v_coeffs, v_dirs};
}
py::tuple sh_bwd(int64_t degree, const Tensor& dirs, const Tensor& coeffs, const OptT& masks, const Tensor& v_colors,
                 bool need_dirs, int64_t stream) {
    ShBwdT o = sh_bwd_c(degree, dirs, coeffs, masks, v_colors, need_dirs, stream);
    return py::make_tuple(o.rc, o.v_coeffs, o.v_dirs);
}

// ---- a9 / a11 ------------------------------------------------------------------------------------------------
// -> (rc, render_colors, render_alphas, last_ids | None)
struct RasterFwdT { int rc; Tensor colors, alphas; OptT last; };
RasterFwdT rasterize_fwd_c(const Tensor& means2d, const Tensor& conics, const Tensor& colors, const Tensor& opacities,
                           const OptT& backgrounds, const OptT& masks, int64_t width, int64_t height, int64_t tile_size,
                           const Tensor& offsets, const Tensor& flatten_ids, bool want_last, const OptT& order,
                           const OptT& work, int64_t stream) {
    req(means2d, at::kFloat, "means2d"); req(conics, at::kFloat, "conics"); req(colors, at::kFloat, "colors");
    req(opacities, at::kFloat, "opacities"); req(offsets, at::kInt, "isect_offsets"); req(flatten_ids, at::kInt, "flatten_ids");
    const int64_t C = opacities.size(0), N = opacities.size(1), D = colors.size(-1);
    const int64_t th = offsets.size(1), tw = offsets.size(2);
    Tensor rc_ = at::empty({C, height, width, D}, f32(means2d));
    Tensor ra = at::empty({C, height, width, 1}, f32(means2d));
    OptT last;
    if (want_last) last = at::empty({C, height, width}, i32(means2d));
    const int rc = sc_rasterize_fwd(fp(means2d), fp(conics), fp(colors), fp(opacities), fpo(backgrounds),
                                    masks ? static_cast<const uint8_t*>(masks->data_ptr()) : nullptr, (int)C, (int)N, (int)D,
                                    (int)width, (int)height, (int)tile_size, (int)tw, (int)th, ip(offsets), ip(flatten_ids),
                                    flatten_ids.numel(), fpw(rc_), fpw(ra),
                                    last ? static_cast<int32_t*>(last->data_ptr()) : nullptr,
                                    order ? ip(*order) : nullptr, work ? static_cast<int32_t*>(work->data_ptr()) : nullptr,
                                    S(stream));
    return {rc, rc_, ra, last};
}
// render_colors as planes [C][D][H][W], handed out as the permuted [C,H,W,D] view (sc_rasterize_fwd_planar; inference only).
// rc == SC_EUNSUPPORTED: nothing was launched, the caller takes rasterize_fwd.
py::tuple rasterize_fwd_planar(const Tensor& means2d, const Tensor& conics, const Tensor& colors, const Tensor& opacities,
                               const OptT& backgrounds, const OptT& masks, int64_t width, int64_t height, int64_t tile_size,
                               const Tensor& offsets, const Tensor& flatten_ids, const OptT& order, const OptT& work,
                               int64_t stream) {
    req(means2d, at::kFloat, "means2d"); req(conics, at::kFloat, "conics"); req(colors, at::kFloat, "colors");
    req(opacities, at::kFloat, "opacities"); req(offsets, at::kInt, "isect_offsets"); req(flatten_ids, at::kInt, "flatten_ids");
    const int64_t C = opacities.size(0), N = opacities.size(1), D = colors.size(-1);
    const int64_t th = offsets.size(1), tw = offsets.size(2);
    Tensor planes = at::empty({C, D, height, width}, f32(means2d));
    Tensor ra = at::empty({C, height, width, 1}, f32(means2d));
    const int rc = sc_rasterize_fwd_planar(fp(means2d), fp(conics), fp(colors), fp(opacities), fpo(backgrounds),
                                           masks ? static_cast<const uint8_t*>(masks->data_ptr()) : nullptr, (int)C, (int)N,
                                           (int)D, (int)width, (int)height, (int)tile_size, (int)tw, (int)th, ip(offsets),
                                           ip(flatten_ids), flatten_ids.numel(), fpw(planes), fpw(ra),
                                           order ? ip(*order) : nullptr,
                                           work ? static_cast<int32_t*>(work->data_ptr()) : nullptr, S(stream));
    return py::make_tuple(rc, planes.permute({0, 2, 3, 1}), ra);
}
py::tuple rasterize_fwd(const Tensor& means2d, const Tensor& conics, const Tensor& colors, const Tensor& opacities,
                        const OptT& backgrounds, const OptT& masks, int64_t width, int64_t height, int64_t tile_size,
                        const Tensor& offsets, const Tensor& flatten_ids, bool want_last, const OptT& order,
                        const OptT& work, int64_t stream) {
    RasterFwdT o = rasterize_fwd_c(means2d, conics, colors, opacities, backgrounds, masks, width, height, tile_size, offsets,
                                   flatten_ids, want_last, order, work, stream);
    return py::make_tuple(o.rc, o.colors, o.alphas, o.last);
}

// ONE zero-filled buffer for the five gradient outputs (the kernel accumulates with float atomics).
// -> (rc, v_means2d, v_conics, v_colors, v_opacities, v_means2d_abs | None)
struct RasterBwdT { int rc; Tensor v_m, v_c, v_col, v_o; OptT v_abs; };
RasterBwdT rasterize_bwd_c(const Tensor& means2d, const Tensor& conics, const Tensor& colors, const Tensor& opacities,
                           const OptT& backgrounds, const OptT& masks, int64_t width, int64_t height, int64_t tile_size,
                           const Tensor& offsets, const Tensor& flatten_ids, const Tensor& render_alphas, const Tensor& last_ids,
                           const Tensor& v_render_colors, const Tensor& v_render_alphas, bool absgrad, const OptT& order,
                           int64_t stream) {
    req(v_render_colors, at::kFloat, "v_render_colors"); req(v_render_alphas, at::kFloat, "v_render_alphas");
    req(last_ids, at::kInt, "last_ids");
    const int64_t C = opacities.size(0), N = opacities.size(1), D = colors.size(-1), CN = C * N;
    const int64_t th = offsets.size(1), tw = offsets.size(2);
    const int64_t sizes[5] = {2 * CN, 3 * CN, D * CN, CN, absgrad ? 2 * CN : 0};
    Tensor flat = at::zeros({sizes[0] + sizes[1] + sizes[2] + sizes[3] + sizes[4]}, f32(means2d));
    int64_t o = 0;
    Tensor v_m = flat.narrow(0, o, sizes[0]).view({C, N, 2}); o += sizes[0];
    Tensor v_c = flat.narrow(0, o, sizes[1]).view({C, N, 3}); o += sizes[1];
    Tensor v_col = flat.narrow(0, o, sizes[2]).view({C, N, D}); o += sizes[2];
    Tensor v_o = flat.narrow(0, o, sizes[3]).view({C, N}); o += sizes[3];
    OptT v_abs;
    if (absgrad) v_abs = flat.narrow(0, o, sizes[4]).view({C, N, 2});
    const int rc = sc_rasterize_bwd(fp(means2d), fp(conics), fp(colors), fp(opacities), fpo(backgrounds),
                                    masks ? static_cast<const uint8_t*>(masks->data_ptr()) : nullptr, (int)C, (int)N, (int)D,
                                    (int)width, (int)height, (int)tile_size, (int)tw, (int)th, ip(offsets), ip(flatten_ids),
                                    flatten_ids.numel(), fp(render_alphas), ip(last_ids), fp(v_render_colors),
                                    fp(v_render_alphas), v_abs ? fpw(*v_abs) : nullptr, fpw(v_m), fpw(v_c), fpw(v_col),
                                    fpw(v_o), order ? ip(*order) : nullptr, S(stream));
    return {rc, v_m, v_c, v_col, v_o, v_abs};
}
py::tuple rasterize_bwd(const Tensor& means2d, const Tensor& conics, const Tensor& colors, const Tensor& opacities,
                        const OptT& backgrounds, const OptT& masks, int64_t width, int64_t height, int64_t tile_size,
                        const Tensor& offsets, const Tensor& flatten_ids, const Tensor& render_alphas, const Tensor& last_ids,
                        const Tensor& v_render_colors, const Tensor& v_render_alphas, bool absgrad, const OptT& order,
                        int64_t stream) {
    RasterBwdT o = rasterize_bwd_c(means2d, conics, colors, opacities, backgrounds, masks, width, height, tile_size, offsets,
                                   flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas, absgrad, order, stream);
    return py::make_tuple(o.rc, o.v_m, o.v_c, o.v_col, o.v_o, o.v_abs);
}

// ---- the fused forward behind gsplat.rendering.rasterization() --------------------------------------
// -> (rc, radii, means2d, depths, records | None, conics | None, opacities | None, colors4 | None)
py::tuple projection_sh_fwd(const Tensor& means, const Tensor& quats, const Tensor& scales, const Tensor& opacities,
                            const Tensor& sh, const Tensor& viewmats, const Tensor& Ks, const Tensor& centers,
                            int64_t sh_degree, int64_t width, int64_t height, double eps2d, double near_plane,
                            double far_plane, double radius_clip, bool antialiased, bool want_records, int64_t stream) {
    req(means, at::kFloat, "means"); req(quats, at::kFloat, "quats"); req(scales, at::kFloat, "scales");
    req(opacities, at::kFloat, "opacities"); req(sh, at::kFloat, "colors"); req(viewmats, at::kFloat, "viewmats");
    req(Ks, at::kFloat, "Ks"); req(centers, at::kFloat, "camera_centers");
    const int64_t C = viewmats.size(0), N = means.size(0), K = sh.size(1);
    Tensor radii = at::empty({C, N}, i32(means));
    Tensor means2d = at::empty({C, N, 2}, f32(means));
    Tensor depths = at::empty({C, N}, f32(means));
    OptT records, conics, opac, cols;
    if (want_records) {
        records = at::empty({C, N, 12}, f32(means));
    } else {
        conics = at::empty({C, N, 3}, f32(means));
        opac = at::empty({C, N}, f32(means));
        cols = at::empty({C, N, 4}, f32(means));
    }
    const int rc = sc_projection_sh_fwd(fp(means), fp(quats), fp(scales), fp(opacities), fp(sh), fp(viewmats), fp(Ks),
                                        fp(centers), (int)C, (int)N, (int)K, (int)sh_degree, (int)width, (int)height,
                                        (float)eps2d, (float)near_plane, (float)far_plane, (float)radius_clip,
                                        antialiased ? 1 : 0, static_cast<int32_t*>(radii.data_ptr()), fpw(means2d),
                                        fpw(depths), conics ? fpw(*conics) : nullptr, opac ? fpw(*opac) : nullptr,
                                        cols ? fpw(*cols) : nullptr, records ? fpw(*records) : nullptr, S(stream));
    return py::make_tuple(rc, radii, means2d, depths, records, conics, opac, cols);
}

// -> (rc, render_colors [C,H,W,4], render_alphas [C,H,W,1])
py::tuple rasterize_fwd_packed(const Tensor& records, const OptT& backgrounds, int64_t width, int64_t height,
                               const Tensor& offsets, const Tensor& flatten_ids, const OptT& order, const OptT& work,
                               bool depth_normalise, int64_t stream) {
    req(records, at::kFloat, "records"); req(offsets, at::kInt, "isect_offsets"); req(flatten_ids, at::kInt, "flatten_ids");
    const int64_t C = records.size(0), N = records.size(1);
    const int64_t th = offsets.size(1), tw = offsets.size(2);
    Tensor rc_ = at::empty({C, height, width, 4}, f32(records));
    Tensor ra = at::empty({C, height, width, 1}, f32(records));
    const int rc = sc_rasterize_fwd_packed(fp(records), fpo(backgrounds), nullptr, (int)C, (int)N, (int)width, (int)height,
                                           (int)tw, (int)th, ip(offsets), ip(flatten_ids), flatten_ids.numel(), fpw(rc_),
                                           fpw(ra), order ? ip(*order) : nullptr,
                                           work ? static_cast<int32_t*>(work->data_ptr()) : nullptr,
                                           depth_normalise ? 1 : 0, S(stream));
    return py::make_tuple(rc, rc_, ra);
}

// ---- the three differentiable operators as C++ autograd functions ------------------------------------------------
// The training step (train.py:236) is bound by HOST time on a busy box (BENCH_r02: 786 steps/s on the driver's box, 1060 on a
// quiet one, same kernels): per step torch.autograd.Function.apply costs ~15 us of Python per operator on the way forward
// and a Python call from the autograd engine's thread (GIL, argument wrapping) per operator on the way back.  Here forward
// and backward are the C++ cores above: no Python frame, and in the backward the GIL is taken only to ask torch for the
// current raw stream (the engine runs a backward on its forward's stream) and for gsplat's `.absgrad` contract (one
// attribute assignment on the caller's tensor object).  The Python autograd.Functions of rendering.py stay: they are the
// ctypes path's, the A/B (`rendering.set_native_autograd(False)`), and what runs while a backward probe is set.
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct PyRef : torch::CustomClassHolder {      // a Python object kept alive by an autograd node
    PyObject* p;
    explicit PyRef(PyObject* o) : p(o) { Py_XINCREF(p); }
    ~PyRef() override {
        if (p && Py_IsInitialized()) { py::gil_scoped_acquire g; Py_DECREF(p); }
    }
};
inline c10::IValue keep(const py::object& o) { return c10::IValue::make_capsule(c10::make_intrusive<PyRef>(o.ptr())); }
inline PyObject* kept(const c10::IValue& v) { return c10::static_intrusive_pointer_cast<PyRef>(v.toCapsule())->p; }
// torch._C._cuda_getCurrentRawStream(device index), asked under the GIL
inline int64_t raw_stream_of(const c10::IValue& stream_fn, const Tensor& like) {
    py::gil_scoped_acquire g;
    return py::reinterpret_borrow<py::object>(kept(stream_fn))((int)like.get_device()).cast<int64_t>();
}
inline void check_rc(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed: ", sc_error_string(rc), " (code ", rc, ")");
}
inline Tensor or_empty(const OptT& t, const Tensor& like) { return t.has_value() ? *t : at::empty({0}, f32(like)); }

struct ProjectionFn : public torch::autograd::Function<ProjectionFn> {
    static variable_list forward(AutogradContext* ctx, const Tensor& means, const Tensor& quats, const Tensor& scales,
                                 const Tensor& viewmats, const Tensor& Ks, int64_t width, int64_t height, double eps2d,
                                 double near_plane, double far_plane, double radius_clip, bool calc_comp,
                                 py::object stream_fn) {
        const c10::IValue sf = keep(stream_fn);
        ProjOutT o = projection_fwd_c(means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                                      radius_clip, calc_comp, raw_stream_of(sf, means));
        check_rc(o.rc, "sc_projection_fwd");
        ctx->save_for_backward({means, quats, scales, viewmats, Ks, o.radii, o.conics, or_empty(o.comps, means)});
        ctx->saved_data["w"] = width; ctx->saved_data["h"] = height; ctx->saved_data["eps"] = eps2d;
        ctx->saved_data["comp"] = calc_comp; ctx->saved_data["sf"] = sf;
        ctx->mark_non_differentiable({o.radii});
        if (calc_comp) return {o.radii, o.means2d, o.depths, o.conics, *o.comps};
        return {o.radii, o.means2d, o.depths, o.conics};
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        const auto sv = ctx->get_saved_variables();
        const Tensor &means = sv[0], &quats = sv[1], &scales = sv[2], &viewmats = sv[3], &Ks = sv[4];
        const bool has_comp = ctx->saved_data["comp"].toBool();
        const int64_t C = viewmats.size(0), N = means.size(0);
        auto z = [&](const Tensor& t, at::IntArrayRef shape) {
            return t.defined() ? t.contiguous() : at::zeros(shape, f32(means));
        };
        const Tensor v_m2 = z(g[1], {C, N, 2}), v_d = z(g[2], {C, N}), v_c = z(g[3], {C, N, 3});
        OptT v_comp, comps;
        if (has_comp) { v_comp = z(g.size() > 4 ? g[4] : Tensor(), {C, N}); comps = sv[7]; }
        ProjBwdT o = projection_bwd_c(means, quats, scales, viewmats, Ks, ctx->saved_data["w"].toInt(),
                                      ctx->saved_data["h"].toInt(), ctx->saved_data["eps"].toDouble(), sv[5], sv[6], comps,
                                      v_m2, v_d, v_c, v_comp, raw_stream_of(ctx->saved_data["sf"], means));
        check_rc(o.rc, "sc_projection_bwd");
        return {ctx->needs_input_grad(0) ? o.v_means : Tensor(), ctx->needs_input_grad(1) ? o.v_quats : Tensor(),
                ctx->needs_input_grad(2) ? o.v_scales : Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

struct ShFn : public torch::autograd::Function<ShFn> {
    static Tensor forward(AutogradContext* ctx, int64_t degree, const Tensor& dirs, const Tensor& coeffs, const OptT& masks,
                          py::object stream_fn) {
        const c10::IValue sf = keep(stream_fn);
        ShFwdT o = sh_fwd_c(degree, dirs, coeffs, masks, raw_stream_of(sf, dirs));
        check_rc(o.rc, "sc_sh_fwd");
        ctx->save_for_backward({dirs, coeffs, or_empty(masks, dirs)});
        ctx->saved_data["deg"] = degree; ctx->saved_data["mask"] = masks.has_value(); ctx->saved_data["sf"] = sf;
        return o.colors;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        const auto sv = ctx->get_saved_variables();
        OptT masks;
        if (ctx->saved_data["mask"].toBool()) masks = sv[2];
        const Tensor v_colors = g[0].defined() ? g[0].contiguous() : at::zeros_like(sv[0]);
        // (needs_input_grad counts the TENSOR inputs of forward: dirs 0, coeffs 1)
        ShBwdT o = sh_bwd_c(ctx->saved_data["deg"].toInt(), sv[0], sv[1], masks, v_colors, ctx->needs_input_grad(0),
                            raw_stream_of(ctx->saved_data["sf"], sv[0]));
        check_rc(o.rc, "sc_sh_bwd");
        return {Tensor(), o.v_dirs.has_value() ? *o.v_dirs : Tensor(), ctx->needs_input_grad(1) ? o.v_coeffs : Tensor(),
                Tensor(), Tensor()};
    }
};

struct RasterFn : public torch::autograd::Function<RasterFn> {
    static variable_list forward(AutogradContext* ctx, const Tensor& means2d, const Tensor& conics, const Tensor& colors,
                                 const Tensor& opacities, const OptT& backgrounds, const OptT& masks, int64_t width,
                                 int64_t height, int64_t tile_size, const Tensor& offsets, const Tensor& flatten_ids,
                                 bool absgrad, const OptT& order, const OptT& work, py::object absgrad_target,
                                 py::object stream_fn) {
        const c10::IValue sf = keep(stream_fn);
        const bool needs_bwd = means2d.requires_grad() || conics.requires_grad() || colors.requires_grad() ||
                               opacities.requires_grad() || (backgrounds.has_value() && backgrounds->requires_grad());
        RasterFwdT o = rasterize_fwd_c(means2d, conics, colors, opacities, backgrounds, masks, width, height, tile_size, offsets,
                                       flatten_ids, needs_bwd, order, work, raw_stream_of(sf, means2d));
        check_rc(o.rc, "sc_rasterize_fwd");
        ctx->save_for_backward({means2d, conics, colors, opacities, or_empty(backgrounds, means2d),
                                masks.has_value() ? *masks : at::empty({0}, u8(means2d)), offsets, flatten_ids, o.alphas,
                                o.last.has_value() ? *o.last : at::empty({0}, i32(means2d)),
                                order.has_value() ? *order : at::empty({0}, i32(means2d))});
        ctx->saved_data["w"] = width; ctx->saved_data["h"] = height; ctx->saved_data["tile"] = tile_size;
        ctx->saved_data["abs"] = absgrad; ctx->saved_data["bg"] = backgrounds.has_value();
        ctx->saved_data["mask"] = masks.has_value(); ctx->saved_data["order"] = order.has_value();
        ctx->saved_data["sf"] = sf; ctx->saved_data["target"] = keep(absgrad_target);
        return {o.colors, o.alphas};
    }
    static variable_list backward(AutogradContext* ctx, variable_list g) {
        const auto sv = ctx->get_saved_variables();
        const Tensor &means2d = sv[0], &colors = sv[2], &render_alphas = sv[8];
        OptT bg, masks, order;
        if (ctx->saved_data["bg"].toBool()) bg = sv[4];
        if (ctx->saved_data["mask"].toBool()) masks = sv[5];
        if (ctx->saved_data["order"].toBool()) order = sv[10];
        const bool absgrad = ctx->saved_data["abs"].toBool();
        const Tensor v_rc = g[0].defined() ? g[0].contiguous()
                                           : at::zeros({render_alphas.size(0), render_alphas.size(1), render_alphas.size(2),
                                                        colors.size(-1)}, f32(means2d));
        const Tensor v_ra = g[1].defined() ? g[1].contiguous() : at::zeros_like(render_alphas);
        RasterBwdT o = rasterize_bwd_c(means2d, sv[1], colors, sv[3], bg, masks, ctx->saved_data["w"].toInt(),
                                       ctx->saved_data["h"].toInt(), ctx->saved_data["tile"].toInt(), sv[6], sv[7], render_alphas,
                                       sv[9], v_rc, v_ra, absgrad, order, raw_stream_of(ctx->saved_data["sf"], means2d));
        check_rc(o.rc, "sc_rasterize_bwd");
        if (absgrad && o.v_abs.has_value()) {
            // gsplat's contract (street_gaussian_model.py:505-506): the tensor OBJECT the caller passed as means2d gets `.absgrad`
            py::gil_scoped_acquire gil;
            py::reinterpret_borrow<py::object>(kept(ctx->saved_data["target"])).attr("absgrad") = py::cast(*o.v_abs);
        }
        Tensor v_bg;
        if (bg.has_value() && ctx->needs_input_grad(4)) v_bg = (v_rc * (1.0 - render_alphas)).sum(at::IntArrayRef({1, 2}));
        return {o.v_m, o.v_c, o.v_col, o.v_o, v_bg, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

py::tuple projection_autograd(const Tensor& means, const Tensor& quats, const Tensor& scales, const Tensor& viewmats,
                              const Tensor& Ks, int64_t width, int64_t height, double eps2d, double near_plane,
                              double far_plane, double radius_clip, bool calc_comp, py::object stream_fn) {
    variable_list o = ProjectionFn::apply(means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                                          radius_clip, calc_comp, stream_fn);
    if (calc_comp) return py::make_tuple(o[0], o[1], o[2], o[3], o[4]);
    return py::make_tuple(o[0], o[1], o[2], o[3]);
}
Tensor sh_autograd(int64_t degree, const Tensor& dirs, const Tensor& coeffs, const OptT& masks, py::object stream_fn) {
    return ShFn::apply(degree, dirs, coeffs, masks, stream_fn);
}
py::tuple rasterize_autograd(const Tensor& means2d, const Tensor& conics, const Tensor& colors, const Tensor& opacities,
                             const OptT& backgrounds, const OptT& masks, int64_t width, int64_t height, int64_t tile_size,
                             const Tensor& offsets, const Tensor& flatten_ids, bool absgrad, const OptT& order,
                             const OptT& work, py::object absgrad_target, py::object stream_fn) {
    variable_list o = RasterFn::apply(means2d, conics, colors, opacities, backgrounds, masks, width, height, tile_size, offsets,
                                      flatten_ids, absgrad, order, work, absgrad_target, stream_fn);
    return py::make_tuple(o[0], o[1]);
}

// ---- frame export ---------------------------------------------------------------------------------------------
int frame_composite_u8_strided(int64_t fg_ptr, int64_t fg_pix, int64_t fg_ch, int64_t acc_ptr, int64_t sky_ptr, int64_t sky_pix,
                               int64_t sky_ch, int64_t n_pixels, int64_t rounding, const Tensor& out, int64_t stream) {
    return sc_frame_composite_u8_strided(reinterpret_cast<const float*>(static_cast<uintptr_t>(fg_ptr)), fg_pix, fg_ch,
                                         reinterpret_cast<const float*>(static_cast<uintptr_t>(acc_ptr)),
                                         reinterpret_cast<const float*>(static_cast<uintptr_t>(sky_ptr)), sky_pix, sky_ch,
                                         n_pixels, (int)rounding, static_cast<uint8_t*>(out.data_ptr()), S(stream));
}
int frame_composite_u8(int64_t fg_ptr, int64_t fg_stride, int64_t acc_ptr, int64_t sky_ptr, int64_t sky_stride,
                       int64_t n_pixels, int64_t rounding, const Tensor& out, int64_t stream) {
    return sc_frame_composite_u8(reinterpret_cast<const float*>(static_cast<uintptr_t>(fg_ptr)), (int)fg_stride,
                                 reinterpret_cast<const float*>(static_cast<uintptr_t>(acc_ptr)),
                                 reinterpret_cast<const float*>(static_cast<uintptr_t>(sky_ptr)), (int)sky_stride, n_pixels,
                                 (int)rounding, static_cast<uint8_t*>(out.data_ptr()), S(stream));
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "street_crafter_amd: compiled Python <-> C-ABI binding layer (allocation + call per operator)";
    // (the header digest THIS binary was compiled against, not the loaded library's: _lib.py compares the two)
    m.def("abi_version", []() { return std::string("street_crafter_amd 0.4.0 (gfx950) abi:" SC_ABI_HASH); });
    m.def("projection_fwd", &projection_fwd);
    m.def("projection_bwd", &projection_bwd);
    m.def("isect_bin_count", &isect_bin_count);
    m.def("isect_bin_sort", &isect_bin_sort);
    m.def("wait_i64", &wait_i64);
    m.def("sh_fwd", &sh_fwd);
    m.def("sh_bwd", &sh_bwd);
    m.def("rasterize_fwd", &rasterize_fwd);
    m.def("rasterize_fwd_planar", &rasterize_fwd_planar);
    m.def("rasterize_bwd", &rasterize_bwd);
    m.def("projection_autograd", &projection_autograd);
    m.def("sh_autograd", &sh_autograd);
    m.def("rasterize_autograd", &rasterize_autograd);
    m.def("projection_sh_fwd", &projection_sh_fwd);
    m.def("rasterize_fwd_packed", &rasterize_fwd_packed);
    m.def("frame_composite_u8", &frame_composite_u8);
    m.def("frame_composite_u8_strided", &frame_composite_u8_strided);
}
