/*
 * street_crafter_amd -- C ABI of the MI355X (gfx950) Gaussian-splat hot path.
 *
 * This is the drop-in boundary: plain pointers + sizes, no torch types.  Every pointer is a
 * DEVICE pointer unless its name ends in _host.  `stream` is a hipStream_t passed as void*.
 * Every entry point returns 0 on success, a hipError_t (>0) from the runtime, or a negative
 * SC_E* code for an argument the library rejects before launching anything.
 *
 * What each entry point replaces (reference = zzz5y/street_crafter, paths under /root/reference):
 *   sc_projection_fwd/bwd   gsplat.rendering.fully_fused_projection
 *                           called at street_gaussian/models/street_gaussian_renderer.py:219-232
 *   sc_isect_*              gsplat.rendering.isect_tiles           renderer.py:243-252
 *   sc_isect_offsets        gsplat.rendering.isect_offset_encode   renderer.py:253
 *   sc_sh_fwd/bwd           gsplat.rendering.spherical_harmonics   renderer.py:259
 *   sc_rasterize_fwd/bwd    gsplat.rendering.rasterize_to_pixels   renderer.py:267-280
 *                           (backward reached from train.py:236; absgrad read at
 *                            street_gaussian/models/street_gaussian_model.py:505-506)
 *   sc_knn3_mean_dist2      simple_knn._C.distCUDA2
 *                           street_gaussian/models/gaussian_model.py:65,
 *                           gaussian_model_actor.py:139, data_processor/utils/render_utils.py:125
 * The CUDA sources of gsplat / simple-knn are not vendored in the reference;
 * semantics follow SURVEY.md Appendix A and are pinned by oracle/ + tests/golden/.
 *
 * Layouts: all float tensors fp32, row-major, innermost dimension contiguous:
 *   means[N,3] quats[N,4](wxyz) scales[N,3] viewmats[C,4,4](world->cam) Ks[C,3,3]
 *   radii i32[C,N]  means2d[C,N,2]  depths[C,N]  conics[C,N,3]  compensations[C,N]
 *   isect_ids i64[I] = (cam << (32+tile_bits)) | (tile << 32) | depth_bits ; flatten_ids i32[I] = cam*N+n
 *   isect_offsets i32[C,tile_h,tile_w] ; colors[C,N,D] ; opacities[C,N]
 *   render_colors[C,H,W,D] render_alphas[C,H,W,1] last_ids i32[C,H,W]
 */
#ifndef STREET_CRAFTER_AMD_H
#define STREET_CRAFTER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sc_stream_t;

#define SC_OK 0
#define SC_EINVAL (-1)      /* bad size / null pointer / unsupported parameter */
#define SC_EWORKSPACE (-2)  /* workspace too small */
#define SC_EUNSUPPORTED (-3)

/* ---- library info ------------------------------------------------------------------- */
const char* sc_version(void);
const char* sc_error_string(int code);
/* compiled-for architecture string, e.g. "gfx950" */
const char* sc_target_arch(void);
/* host-side spin until *addr == value (acquire); returns 0, or 1 after timeout_us microseconds (< 0: never).
 * For the sequence number sc_isect_bin_count publishes into host-mapped memory (meta_mirror[4]); a ctypes host
 * calls it without holding the GIL, so other host threads keep launching meanwhile. */
int sc_wait_i64(const int64_t* addr, int64_t value, int64_t timeout_us);

/* ---- a1: projection (renderer.py:219-232) -------------------------------------------- */
int sc_projection_fwd(const float* means, const float* quats, const float* scales,
                      const float* viewmats, const float* Ks, int C, int N, int width, int height,
                      float eps2d, float near_plane, float far_plane, float radius_clip,
                      int32_t* radii, float* means2d, float* depths, float* conics,
                      float* compensations /* nullable */, sc_stream_t stream);

/* VJP of sc_projection_fwd w.r.t. means/quats/scales (camera tensors carry no grad at the
 * reference's call site).  v_compensations nullable.  Outputs are OVERWRITTEN (summed over C). */
int sc_projection_bwd(const float* means, const float* quats, const float* scales,
                      const float* viewmats, const float* Ks, int C, int N, int width, int height,
                      float eps2d, const int32_t* radii, const float* conics,
                      const float* compensations /* nullable */,
                      const float* v_means2d, const float* v_depths, const float* v_conics,
                      const float* v_compensations /* nullable */,
                      float* v_means, float* v_quats, float* v_scales, sc_stream_t stream);

/* ---- a3: tile intersection (renderer.py:243-252) -------------------------------------- */
/* workspace bytes needed by sc_isect_count/sc_isect_emit for CN = C*N gaussians */
size_t sc_isect_workspace_bytes(int64_t CN);
/* pass 1: tiles_per_gauss[C,N] and the total number of intersections (device int64). */
int sc_isect_count(const float* means2d, const int32_t* radii, int C, int N,
                   int tile_size, int tile_width, int tile_height,
                   int32_t* tiles_per_gauss, int64_t* total_dev, void* workspace, size_t ws_bytes,
                   sc_stream_t stream);
/* pass 2: unsorted keys/values in emission order (gaussian-major, row-major over the rect).
 * Must be called with the workspace left by sc_isect_count. */
int sc_isect_emit(const float* means2d, const int32_t* radii, const float* depths, int C, int N,
                  int tile_size, int tile_width, int tile_height,
                  const int32_t* tiles_per_gauss, int64_t n_isects,
                  int64_t* isect_ids, int32_t* flatten_ids, void* workspace, size_t ws_bytes,
                  sc_stream_t stream);

/* stable LSD radix sort of (u64 key, i32 value) pairs over key bits [0, end_bit). In place:
 * on return keys/vals hold the sorted result.  tmp_* are scratch of the same size. */
size_t sc_radix_sort_workspace_bytes(int64_t n);
int sc_radix_sort_pairs_u64_i32(uint64_t* keys, int32_t* vals, uint64_t* tmp_keys, int32_t* tmp_vals,
                                int64_t n, int end_bit, void* workspace, size_t ws_bytes,
                                sc_stream_t stream);

/* Fused tile-bucketed path: produces exactly what count + emit + stable sort + offset-encode
 * produce (bit-identical isect_ids / flatten_ids / offsets): Gaussians are bucketed by SUPER-TILE
 * (2x2 tiles), each super-tile is sorted once in LDS and its up-to-4 per-tile lists are emitted by
 * a stable filter.  Two calls because the caller must size the outputs in between:
 *   sc_isect_bin_count : tiles_per_gauss, per-tile offsets (= isect_offset_encode result) and
 *                        meta_dev[0] = total intersections I, [1] = largest per-tile count,
 *                        [2] = total (Gaussian, super-tile) records, [3] = largest super-tile
 *   sc_isect_bin_sort  : isect_ids / flatten_ids, sorted.  `capacity` = elements the output buffers
 *                        were sized for, `rec_capacity` = records the workspace was sized for,
 *                        `super_capacity` = largest super-tile the caller provisioned LDS for.  The
 *                        kernels read meta_dev themselves and do NOTHING when meta_dev[0] > capacity,
 *                        meta_dev[2] > rec_capacity or meta_dev[3] > super_capacity, so a caller may
 *                        launch with predicted sizes before it has read meta_dev back (no GPU idle
 *                        bubble) and retry with exact sizes if the prediction was too small.
 *                        A super-tile bucket of up to 3584 records is sorted by one workgroup in LDS; when
 *                        super_capacity is larger, longer buckets are first cut into depth ranges that fit
 *                        (only they pay for it).
 * Both return SC_EUNSUPPORTED when C*tile_width*tile_height > 36864 (a 3840x2160 frame has 32400 tiles), C*N >= 2^28 or
 * super_capacity > 220 000; the caller then takes the count/emit/radix-sort route.
 */
/* n_records < 0: bytes of the count-phase workspace (shared by both calls; holds the visible Gaussians'
 * rectangle / depth / id, 16 B each, in spatial order); n_records >= 0: bytes of the sort-phase workspace for that many records. */
size_t sc_isect_bin_workspace_bytes(int64_t CN, int C, int tile_width, int tile_height, int64_t n_records);
/* meta_mirror (nullable): HOST-MAPPED pinned int64[5] (hipHostMalloc; torch pin_memory).  When given,
 * the device stores meta[0..3] there and then `seq` into meta_mirror[4] with system-scope release, so
 * the host can poll meta_mirror[4] == seq instead of enqueuing a D2H copy + event (which costs a copy
 * kernel and a barrier bubble in the middle of the frame). */
int sc_isect_bin_count(const float* means2d, const int32_t* radii, const float* depths, int C, int N,
                       int tile_size, int tile_width, int tile_height,
                       int32_t* tiles_per_gauss, int32_t* isect_offsets, int64_t* meta_dev /* [4] */,
                       int64_t* meta_mirror /* [5], nullable */, int64_t seq,
                       void* workspace, size_t ws_bytes,
                       const int32_t* tile_work /* nullable: [sc_view_slots()][C*tile_width*tile_height], see
                           sc_rasterize_fwd */,
                       const float* viewmats /* nullable (= slot 0): [C,4,4] world->camera, the frame's cameras: which
                           bank of tile_work this call uses is looked up from camera 0's forward axis, see VIEW SLOTS */,
                       int32_t* view_registry /* nullable (= slot 0): int32[sc_view_registry_words()], VIEW SLOTS */,
                       int32_t* tile_order /* nullable: out, the rasterizer's dispatch list built from tile_work:
                           sc_tile_order_len(C*tile_width*tile_height) items, see sc_rasterize_fwd */,
                       sc_stream_t stream);
/* VIEW SLOTS.  The rasterizer's work hint (tile_work) only helps the frame that finds it if it was left by the
 * same view; a street rig renders front / front-left / front-right in turn.  tile_work therefore has sc_view_slots()
 * banks of C*tile_width*tile_height words, and sc_isect_bin_count chooses the bank on the DEVICE (no host round
 * trip, no extra launch): camera 0's forward axis (row 2 of the world->camera rotation of viewmats [C,4,4]) is
 * matched against `view_registry` (persistent, caller-owned, zero-initialised int32[sc_view_registry_words()], one
 * per device); a slot within ~7 degrees is reused and follows the camera, otherwise the least recently used one
 * is taken over.  The slot is stored in tile_order's last word, where the rasterizer finds the bank to report
 * into.  Scheduling only: any slot renders the same image. */
int sc_view_slots(void);
int sc_view_registry_words(void);
/* Records of the largest super-tile bucket one workgroup of sc_isect_bin_sort sorts in LDS (3584).  A caller that
 * predicts `super_capacity` with head-room should not let the head-room alone cross this value: above it every call
 * also launches the split kernel and the segment workgroups, whether or not a bucket is that large. */
int sc_isect_bin_bucket_capacity(void);
int sc_isect_bin_sort(const float* means2d, const int32_t* radii, const float* depths, int C, int N,
                      int tile_size, int tile_width, int tile_height,
                      const int32_t* isect_offsets, const int64_t* meta_dev,
                      void* count_workspace /* the one sc_isect_bin_count filled: its scratch counters are
                                               consumed by the first launch whose capacities suffice */,
                      int64_t capacity, int64_t rec_capacity, int64_t super_capacity,
                      int64_t* isect_ids /* nullable */, int32_t* flatten_ids,
                      void* workspace, size_t ws_bytes, sc_stream_t stream);
/* isect_ids from the sorted lists: isect_ids[k] = (camera << (32 + tile_bits)) | (tile << 32) | bits(depths[
 * flatten_ids[k]]) for every k in tile's range of isect_offsets.  Lets a caller skip the 8 B x I key array in
 * sc_isect_bin_sort (isect_ids = NULL) and produce it only if somebody asks for it. */
int sc_isect_ids_rebuild(const int32_t* flatten_ids, const int32_t* isect_offsets, const float* depths,
                         int C, int N, int tile_width, int tile_height, int64_t n_isects,
                         int64_t* isect_ids, sc_stream_t stream);
/* Re-zero the sort phase's bucket cursors / fallback flags inside `count_workspace`: call before a
 * SECOND sc_isect_bin_sort of the same count phase (retry after an under-predicted capacity). */
int sc_isect_bin_reset_cursors(void* count_workspace, int64_t CN, int C, int tile_width, int tile_height,
                               sc_stream_t stream);

/* ---- a4: offsets (renderer.py:253) ---------------------------------------------------- */
int sc_isect_offsets(const int64_t* isect_ids, int64_t n_isects, int C, int tile_width,
                     int tile_height, int32_t* offsets, sc_stream_t stream);

/* ---- a6: spherical harmonics (renderer.py:259) ---------------------------------------- */
/* M rows; dirs[M,3], coeffs[M,K,3] (K >= (degree+1)^2 coefficients per row, row stride K),
 * masks uint8[M] nullable; colors[M,3].  Masked-out rows are written as 0. */
int sc_sh_fwd(int degree, const float* dirs, const float* coeffs, const uint8_t* masks,
              int64_t M, int K, float* colors, sc_stream_t stream);
/* v_dirs nullable. v_coeffs[M,K,3] fully written (zeros beyond the used bases / masked rows). */
int sc_sh_bwd(int degree, const float* dirs, const float* coeffs, const uint8_t* masks,
              int64_t M, int K, const float* v_colors, float* v_coeffs, float* v_dirs,
              sc_stream_t stream);

/* ---- a9: rasterize (renderer.py:267-280) ---------------------------------------------- */
/* D = number of colour channels, 1..32 (3 and 4 have dedicated kernels).
 * backgrounds[C,D] nullable; tile_masks uint8[C,tile_h,tile_w] nullable (0 = skip tile). */
int sc_rasterize_fwd(const float* means2d, const float* conics, const float* colors,
                     const float* opacities, const float* backgrounds, const uint8_t* tile_masks,
                     int C, int N, int D, int width, int height, int tile_size,
                     int tile_width, int tile_height,
                     const int32_t* isect_offsets, const int32_t* flatten_ids, int64_t n_isects,
                     float* render_colors, float* render_alphas,
                     int32_t* last_ids /* nullable: only the backward pass reads it */,
                     const int32_t* tile_order /* nullable: the DISPATCH LIST buffer (sc_tile_order_len(C*tile_width*
                         tile_height) items; the forward reads its leading part), one item per workgroup in launch order (longest-running first: the launch's makespan is one
                         tile's serial walk plus the throughput part).  item = flat tile << 2 | kind: kind 0 = the whole
                         tile, 1 / 2 = its upper / lower 16 x 8 half (a tile whose walk would be the launch's tail is
                         shared by two waves); negative = no work.  Every tile must appear exactly once as kind 0 or
                         once as each of kinds 1 and 2 (a tile that does not appear is not rendered).
                         sc_isect_bin_count builds it */,
                     int32_t* tile_work /* nullable: [sc_view_slots()][C*tile_width*tile_height]; the bank named by
                         tile_order's last word (0..sc_view_slots()-1, clamped) receives the list entries every tile
                         walked: the scheduling hint the NEXT frame of that view has its tile_order built from
                         (persistent, caller-owned, zero-initialised; stale or half-updated values are fine).  Without
                         tile_order: bank 0 */,
                     sc_stream_t stream);
/* number of int32 items in the dispatch-list buffer for `total_tiles` tiles: the forward's list (total_tiles +
 * total_tiles / 8 + 8 items: every tile + room for the split ones, padded with -1), then a whole-tile list in the
 * same order (total_tiles items `tile << 2`), one word that says whether that second list was built: it is only
 * under sc_set_option raster_bwd_split 0 (set BEFORE the frame's sc_isect_bin_count), for a backward that takes
 * whole tiles; by default sc_rasterize_bwd follows the forward's list, half tiles included -- and, last, the view
 * slot of the call (see VIEW SLOTS). */
int sc_tile_order_len(int total_tiles);
/* Gradient outputs must be ZERO-FILLED by the caller (the kernel accumulates with atomics).
 * v_means2d_abs nullable (absgrad). */
int sc_rasterize_bwd(const float* means2d, const float* conics, const float* colors,
                     const float* opacities, const float* backgrounds, const uint8_t* tile_masks,
                     int C, int N, int D, int width, int height, int tile_size,
                     int tile_width, int tile_height,
                     const int32_t* isect_offsets, const int32_t* flatten_ids, int64_t n_isects,
                     const float* render_alphas, const int32_t* last_ids,
                     const float* v_render_colors, const float* v_render_alphas,
                     float* v_means2d_abs, float* v_means2d, float* v_conics, float* v_colors,
                     float* v_opacities, const int32_t* tile_order /* nullable, as sc_rasterize_fwd */,
                     sc_stream_t stream);

/* ---- a14: simple_knn.distCUDA2 --------------------------------------------------------- */
size_t sc_knn_workspace_bytes(int64_t n);
int sc_knn3_mean_dist2(const float* points, int64_t n, float* out, void* workspace,
                       size_t ws_bytes, sc_stream_t stream);

/* ---- fused forward behind gsplat.rendering.rasterization() (imported at
 *      street_gaussian/models/street_gaussian_renderer.py:204) -------------------------------------
 * sc_camera_centers: out[c] = -R^T t of the rigid world-to-camera matrices viewmats [C,4,4]
 *   (= Camera.camera_center, street_gaussian/utils/camera_utils.py:51; gsplat takes inverse(viewmat)).
 * sc_projection_sh_fwd: renderer.py:219-266 in one pass per (camera, Gaussian): projection, opacity *
 *   compensation (when antialiased), dirs = mean - camera centre, SH colour where radius > 0,
 *   clamp_min(colour + 0.5, 0), depth appended as 4th channel.  opacities [N], sh_coeffs [N,K,3];
 *   outputs radii [C,N], means2d [C,N,2], depths [C,N], conics [C,N,3], opacities_out [C,N],
 *   colors4 [C,N,4].  Bit-identical to the separate operators + torch glue.
 * sc_rasterize_fwd_ed: sc_rasterize_fwd (no last_ids) whose 4th output channel is divided by
 *   max(alpha, 1e-10) (renderer.py:284; gsplat render_mode "RGB+ED").  D must be 4 and tile_size 16,
 *   otherwise SC_EUNSUPPORTED. */
int sc_camera_centers(const float* viewmats, int C, float* out, sc_stream_t stream);
int sc_projection_sh_fwd(const float* means, const float* quats, const float* scales,
                         const float* opacities, const float* sh_coeffs, const float* viewmats,
                         const float* Ks, const float* camera_centers, int C, int N, int K,
                         int sh_degree, int width, int height, float eps2d, float near_plane,
                         float far_plane, float radius_clip, int antialiased, int32_t* radii,
                         float* means2d, float* depths, float* conics, float* opacities_out,
                         float* colors4,
                         float* records /* nullable: [C,N,12], 16-B aligned: everything the rasterizer gathers per
                             splat in one 48-B record (x, y, conic a, b | conic c, opacity, colour 0, 1 | colour 2,
                             3, -, -) for sc_rasterize_fwd_packed.  With records, conics / opacities_out / colors4
                             may each be NULL (sc_records_unpack rebuilds them on demand) */,
                         sc_stream_t stream);
/* sc_rasterize_fwd (4 channels, tile 16, no last_ids) reading `records` instead of the four parameter arrays: one
 * gather line per splat instead of four.  depth_normalise != 0: the epilogue of
 * sc_rasterize_fwd_ed.  Bit-identical output.  SC_EUNSUPPORTED when the reference-shaped raster kernel is selected. */
int sc_rasterize_fwd_packed(const float* records, const float* backgrounds, const uint8_t* tile_masks,
                            int C, int N, int width, int height, int tile_width, int tile_height,
                            const int32_t* isect_offsets, const int32_t* flatten_ids, int64_t n_isects,
                            float* render_colors, float* render_alphas, const int32_t* tile_order,
                            int32_t* tile_work, int depth_normalise, sc_stream_t stream);
/* sc_rasterize_fwd with render_colors stored as PLANES, [C][D][H][W] (one plane per channel), for inference (no
 * last_ids).  The reference's caller slices the result channel-wise right behind the operator
 * (street_gaussian_renderer.py:282-300: `render_colors[..., :-1]`, `[..., -1:] / alpha`, clamp, `.permute(2, 0, 1)`):
 * strided kernels over the interleaved buffer, dense ones over planes.  The Python operator hands the buffer out as a
 * permuted [C,H,W,D] view, so the values a caller sees are identical.  Wave kernel only (tile_size 16, D = 3 or 4):
 * SC_EUNSUPPORTED otherwise. */
int sc_rasterize_fwd_planar(const float* means2d, const float* conics, const float* colors,
                            const float* opacities, const float* backgrounds, const uint8_t* tile_masks,
                            int C, int N, int D, int width, int height, int tile_size,
                            int tile_width, int tile_height,
                            const int32_t* isect_offsets, const int32_t* flatten_ids, int64_t n_isects,
                            float* render_colors, float* render_alphas,
                            const int32_t* tile_order, int32_t* tile_work, sc_stream_t stream);
/* conics [CN,3] / opacities [CN] / colors4 [CN,4] (each nullable) out of CN records */
int sc_records_unpack(const float* records, int64_t CN, float* conics, float* opacities, float* colors4,
                      sc_stream_t stream);
int sc_rasterize_fwd_ed(const float* means2d, const float* conics, const float* colors,
                        const float* opacities, const float* backgrounds, const uint8_t* tile_masks,
                        int C, int N, int D, int width, int height, int tile_size, int tile_width,
                        int tile_height, const int32_t* isect_offsets, const int32_t* flatten_ids,
                        int64_t n_isects, float* render_colors, float* render_alphas,
                        const int32_t* tile_order, int32_t* tile_work, sc_stream_t stream);

/* ---- frame export for the multi-GPU gather: the tail of render_novel_view
 *      (street_gaussian/models/street_gaussian_renderer.py:151-163: fg + sky * (1 - acc), clamp) and the
 *      visualizer's uint8 conversion (street_gaussian/visualizers/street_gaussian_visualizer.py:88-101),
 *      which the reference does with torch elementwise ops + .cpu().numpy() per frame.
 * fg / sky: the rasterizer's raw f32 images with `*_stride` floats per pixel (>= 3; 4 for RGB+depth);
 * acc: the foreground pass's alpha [n_pixels]; sky and acc are both given or both NULL (single pass).
 * out u8[n_pixels,3] = q(clamp(clamp(fg,0,1) + clamp(sky,0,1) * (1 - acc), 0, 1)), with
 * rounding 0: q(x) = (uint8)(x * 255)        -- the novel-view video frames, `(rgb * 255).astype(np.uint8)`
 * rounding 1: q(x) = (uint8)(x * 255 + 0.5)  -- torchvision.utils.save_image PNGs.
 * Each product / sum is rounded separately: bit-identical to the reference's torch composition. */
int sc_frame_composite_u8(const float* fg, int fg_stride, const float* acc, const float* sky, int sky_stride,
                          int64_t n_pixels, int rounding, uint8_t* out, sc_stream_t stream);
/* The same for images with a channel stride: channel c of pixel i at fg[i * fg_pix_stride + c * fg_ch_stride] -- the
 * planes sc_rasterize_fwd_planar writes have pixel stride 1 and channel stride H * W. */
int sc_frame_composite_u8_strided(const float* fg, int64_t fg_pix_stride, int64_t fg_ch_stride, const float* acc,
                                  const float* sky, int64_t sky_pix_stride, int64_t sky_ch_stride, int64_t n_pixels,
                                  int rounding, uint8_t* out, sc_stream_t stream);

/* unit-test hook for the backward kernel's transposing reduction (v_permlane32/16_swap + DPP):
 * in [n_waves][16][64] per-lane partial sums, out [n_waves][64]: lane l = 64-lane total of value l >> 2 */
int sc_test_wave_transpose_sum16(const float* in, int n_waves, float* out, sc_stream_t stream);

/* ---- HIP streams with a CU mask or a priority (frame loop: street_crafter_amd/dist.py, bench.py) ----------------
 * Frames are independent (render.py:64-70) and several are kept in flight on different streams; the VALU-bound
 * rasterizer (10.7 k one-wave workgroups) otherwise starves the 8..16-wave workgroups of the NEXT frame's latency-bound
 * intersection kernels of CU slots.  sc_stream_create makes a non-blocking stream that is either confined to a set of
 * CUs (n_mask_words > 0: hipExtStreamCreateWithCUMask; bit i of the mask = CU i, the driver deals the bits round-robin
 * over the 8 XCDs, so "the first K bits" is K / 8 CUs of every XCD) or has a priority (n_mask_words == 0:
 * hipStreamCreateWithPriority; the range is returned by sc_stream_priority_range, lower number = higher priority).
 * The handle is a hipStream_t; the caller destroys it with sc_stream_destroy after synchronising it. */
int sc_stream_create(int priority, const uint32_t* cu_mask, int n_mask_words, sc_stream_t* out);
int sc_stream_destroy(sc_stream_t stream);
int sc_stream_priority_range(int* least, int* greatest);

/* ---- tuning / introspection ------------------------------------------------------------ */
/* Select a kernel variant at run time (for A/B measurements in one process).
 *   key "proj_clamp": the clamp of x/z, y/z in fully_fused_projection's EWA Jacobian (forward, fused forward and backward):
 *                     0 = +-1.3 tan(fov/2) (upstream gsplat v1.0-1.3; default), 1 = [-(cx/fx + 0.3 tan), (W - cx)/fx +
 *                     0.3 tan] (v1.4+).  The two agree for a centred principal point.
 *   key "radius_floor": the floor under the discriminant of the 3-sigma radius, sqrt(max(floor, b^2 - det)):
 *                     0 = 0.01 (gsplat v1.x; default), 1 = 0.1 (the original Inria rasterizer and early forks).
 *                     The reference installs an UNPINNED gsplat fork (README.md:35): INTEGRATION.md says how to tell which
 *                     pair a checkout has.  Environment: SC_PROJ_CLAMP=asymmetric / SC_RADIUS_FLOOR=0.1 set them at load.
 *   key "isect_pull": 1 = sc_isect_bin_count / _sort take the PULL route (every super-tile bucket's sort workgroup
 *                     gathers its own records from a (size class, anchor)-sorted payload: no scatter launch, no records
 *                     buffer) for frames whose key table fits the LDS; 0 = the scatter route (default; the pull route
 *                     measured slower, profiles/r04_isect_pull_ab.txt).  Same results bit for bit.  Set it between
 *                     frames (both calls of a frame must see the same value).
 *   key "raster_fwd": 0 = reference-shaped (all pixels x all splats; generic fallback / cross-check),
 *                     3 = one wave per tile, 4 pixels per lane, exact tile-level cull (default)
 *   key "raster_map": block -> tile map of the wave rasterizer: 1 = neighbouring tiles round-robin over the
 *                     8 XCDs (default), 0 = one band of tile rows per XCD
 *   key "raster_split": 0..100: the dispatch list sc_isect_bin_count builds lists a tile as two 16 x 8 halves when
 *                     its work hint is at least this percentage of the heaviest tile's (default 50; 0 = never)
 *   key "raster_hint_blend": 0..4: a tile's work hint = max(its own, this many quarters of the largest hint within
 *                     2 tiles of it) (default 3; 0 = own value only: exact for a camera that stands still)
 *   key "raster_bwd": 0 = reference-shaped (one lane per pixel), 1 = one wave per tile (default)
 *   key "raster_bwd_split": 1 = the backward follows the forward's dispatch list including its half tiles (default),
 *                     0 = the whole-tile list behind it
 * Returns the previous value, or SC_EINVAL for an unknown key.
 * (The diagnostic skips "debug0".."debug3" of rounds 1-2 are NOT part of this library any more: they exist only in
 *  the separate diagnostic build, lib/libstreet_crafter_hip_diag.so (-DSC_DIAG), which tools/exp_*.py load
 *  explicitly; here they are unknown keys.) */
int sc_set_option(const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif /* STREET_CRAFTER_AMD_H */
